"""CPU oracle of the composite "3-level GNN" benchmark model (``mlgnn.workload.ThreeLevelGNN``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Executes the reference's literal op sequence:
materialised ``[E, d]`` edge embeddings through both Linear encoders, gather -> elementwise ->
scatter aggregation (``oracle.gcn_lib.genconv``), ``[B, G, C, k]`` projection pooling with
``scatter_reduce`` and dense DiffPool -- this is the "reference-semantics CPU path" that
``bench.py`` times as ``cpu_baseline`` (kind "port").
"""
import torch
import torch.nn.functional as F

from . import gcn_lib as G
from . import models as M

N_PATHWAYS, N_GROUPS = 146, 3


def three_level_forward(sd, batch, num_layers=3, aggr="softmax", t=1.0, learn_t=False, msg_norm=False,
                        pool_layers=2):
    x, ei = batch.x, batch.edge_index
    h = F.linear(x, sd["node_features_encoder.weight"], sd["node_features_encoder.bias"])
    edge_emb = F.linear(batch.edge_attr, sd["edge_encoder.weight"], sd["edge_encoder.bias"])     # [E, H]

    def conv(l, inp):
        return G.genconv(inp, ei, edge_emb, sd, "gcns.%d." % l, aggr=aggr, t=t, learn_t=learn_t,
                         msg_norm_on=msg_norm, encode_edge=True, norm_kind="layer", mlp_layers=2)

    def nrm(l, inp):
        return G.norm(inp, "layer", sd, "norms.%d." % l)

    h = conv(0, h)
    for l in range(1, num_layers):
        h = conv(l, F.relu(nrm(l - 1, h))) + h
    h = nrm(num_layers - 1, h)
    B = batch.gene_pca_match.shape[0]
    hidden = h.shape[1]
    k = sd["learnable_pca_params"].shape[1]
    p = M.projection_pool(h, batch.gene_pca_match, batch.raw_indice, sd["learnable_pca_params"], None,
                          batch.nodes_per_graph, N_PATHWAYS * N_GROUPS, True)
    p = p.reshape(B, hidden, N_PATHWAYS, N_GROUPS * k)
    z = p.permute(0, 3, 2, 1).reshape(-1, N_PATHWAYS, hidden)
    z, link, ent = M.diffpool_forward(sd, z, sd["pathway_adj"], pool_layers, 1, prefix="diff_pooling.")
    logits = F.linear(z.reshape(B, -1), sd["head.weight"], sd["head.bias"])
    return F.softmax(logits, dim=-1), link, ent


def training_loss(sd, batch, **kw):
    pred, link, ent = three_level_forward(sd, batch, **kw)
    return F.binary_cross_entropy(pred, batch.y.reshape(-1, 2)) + link + ent
