"""Synthetic workloads of BASELINE.json (no dataset ships with the reference)
and the composite "3-level GNN" the headline metric is quoted on.

Config 2: per graph a directed Erdos-Renyi multigraph ``G(N=10 000, E=160 000)`` (i.i.d. uniform
endpoints, duplicates kept, self loops re-drawn, seed ``1000 + graph_id``), node input
``x ~ N(0,1) [N,3]``, scalar edge attribute ``a ~ U(0,1) [E,1]``; model = DeeperGCN-shaped GENConv
trunk (level 0: gene graph) -> gene->pathway projection pooling to 146 pathways x 3 groups
(level 1) -> 2-level DiffPool 146 -> 37 -> 10 (levels 2-3) -> linear head.  The three stages are
the reference's own hierarchy: ``deepergcn.py:232-247`` (res+ GENConv stack),
``multilevel_gnn.py:212-242`` (projection pooling) and ``vae.py:238-243`` (the DiffPool call site).
"""
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

from .dense import linear
from .graph import CSRGraph
from .ops import RankOneEdge
from .project import segment_project

N_PATHWAYS, N_GROUPS = 146, 3


def er_graph(graph_id, n_nodes, n_edges):
    """One synthetic graph on the CPU: ``(edge_index [2,E] int64, x [N,3], edge_attr [E,1])``."""
    gen = torch.Generator().manual_seed(1000 + int(graph_id))
    src = torch.randint(0, n_nodes, (n_edges,), generator=gen)
    dst = torch.randint(0, n_nodes, (n_edges,), generator=gen)
    loops = src == dst
    while n_nodes > 1 and bool(loops.any()):
        dst[loops] = torch.randint(0, n_nodes, (int(loops.sum()),), generator=gen)
        loops = src == dst
    x = torch.randn(n_nodes, 3, generator=gen)
    ea = torch.rand(n_edges, 1, generator=gen)
    y = int(torch.randint(0, 2, (1,), generator=gen))
    return torch.stack([src, dst]), x, ea, y


def membership(n_nodes, n_members, seed=7):
    """Gene -> (pathway, group) membership table shared by every graph: sorted segment ids
    ``raw_indice [G]`` over 438 segments and node index ``match [G]`` (about 1 % absent = -1)."""
    gen = torch.Generator().manual_seed(seed)
    seg = torch.sort(torch.randint(0, N_PATHWAYS * N_GROUPS, (n_members,), generator=gen))[0]
    match = torch.randint(0, n_nodes, (n_members,), generator=gen)
    match[torch.rand(n_members, generator=gen) < 0.01] = -1
    return match, seg


def _rows(table, B, device):
    """The per-fold membership table as the ``[B, G]`` rows a batch carries.  A table that already lives on the target
    device is expanded (every batch then views the SAME storage, as a loader holding one table per fold would, and
    :func:`mlgnn.project.membership_tables` recognises it); otherwise each batch gets its own copy."""
    if table.device == torch.device(device):
        return table[None, :].expand(B, -1)
    return table[None, :].repeat(B, 1)


def collate(graph_ids, n_nodes, n_edges, match, seg, device="cpu"):
    """PyG-style collate: block-diagonal batch with ``edge_index`` offset by the cumulative node
    count, ``batch`` vector, per-graph ``gene_pca_match``/``raw_indice`` rows (not offset)."""
    eis, xs, eas, ys = [], [], [], []
    for k, gid in enumerate(graph_ids):
        ei, x, ea, y = er_graph(gid, n_nodes, n_edges)
        eis.append(ei + k * n_nodes)
        xs.append(x)
        eas.append(ea)
        ys.append(y)
    B = len(graph_ids)
    labels = F.one_hot(torch.tensor(ys), 2).to(torch.float32)
    b = SimpleNamespace(
        x=torch.cat(xs), edge_index=torch.cat(eis, dim=1), edge_attr=torch.cat(eas),
        batch=torch.arange(B).repeat_interleave(n_nodes), y=labels.reshape(-1),
        gene_pca_match=_rows(match, B, device), raw_indice=_rows(seg, B, device),
        num_graphs=B, nodes_per_graph=n_nodes)
    for k, v in list(vars(b).items()):
        if torch.is_tensor(v):
            setattr(b, k, v.to(device))
    return b


def pathway_adjacency(seed=11):
    gen = torch.Generator().manual_seed(seed)
    a = torch.rand(N_PATHWAYS, N_PATHWAYS, generator=gen)
    return (a + a.t()) / 2 + torch.eye(N_PATHWAYS)


class ThreeLevelGNN(nn.Module):
    """GENConv trunk -> projection pooling -> DiffPool -> head (see module docstring)."""

    def __init__(self, hidden=128, num_layers=3, aggr="softmax", n_members=25000, pca_dim=2,
                 pool_hidden=32, pool_out=64, pool_layers=2, t=1.0, learn_t=False, msg_norm=False):
        super().__init__()
        from models.diff_pooling import DiffPool
        from models.gcn_lib.sparse.torch_nn import norm_layer
        from models.gcn_lib.sparse.torch_vertex import GENConv
        self.hidden, self.num_layers, self.pca_dim = hidden, num_layers, pca_dim
        self.node_features_encoder = nn.Linear(3, hidden)
        self.edge_encoder = nn.Linear(1, hidden)
        self.gcns = nn.ModuleList([GENConv(hidden, hidden, aggr=aggr, t=t, learn_t=learn_t, msg_norm=msg_norm,
                                           encode_edge=True, edge_feat_dim=hidden, norm="layer", mlp_layers=2)
                                   for _ in range(num_layers)])
        self.norms = nn.ModuleList([norm_layer("layer", hidden) for _ in range(num_layers)])
        self.learnable_pca_params = nn.Parameter(torch.randn(n_members, pca_dim) * 0.05)
        self.diff_pooling = DiffPool(hidden, None, N_PATHWAYS, pool_layers, pool_hidden, pool_out,
                                     SimpleNamespace(pooling_type="correlation", after_pooling_layer=1))
        clusters = N_PATHWAYS
        for _ in range(pool_layers):
            clusters = -(-clusters // 4) if pool_layers > 1 else -(-clusters // 10)
        self.head = nn.Linear(pool_out * clusters * N_GROUPS * pca_dim, 2)
        self.register_buffer("pathway_adj", pathway_adjacency())

    def forward(self, batch):
        """-> ``(probabilities [B,2], link_loss, entropy_loss)``."""
        N = batch.x.shape[0]
        graph = getattr(batch, "csr", None)
        if graph is None:
            graph = CSRGraph(batch.edge_index, N)
        h = linear(batch.x, self.node_features_encoder.weight, self.node_features_encoder.bias)
        edge = RankOneEdge(batch.edge_attr[:, 0], self.edge_encoder.weight[:, 0], self.edge_encoder.bias)
        # res+ block (deepergcn.py:232-247), dropout 0: h = conv(relu(norm(h))) + h.  The add, and the norm (+ ReLU) that
        # follows -- the next block's input, the final norm after the last conv -- run in the conv's last GEMM epilogue;
        # the gradient of the identity branch is added inside that LayerNorm's backward
        L = self.num_layers
        h, y = self.gcns[0](h, graph, edge, post_norm=(self.norms[0], L > 1))
        for l in range(1, L):
            h, y = self.gcns[l](y, graph, edge, residual=h, post_norm=(self.norms[l], l < L - 1))
        h = y
        # level 1: gene -> pathway projection pooling (multilevel_gnn.py:212-242)
        B = batch.gene_pca_match.shape[0]
        # levels 2-3: DiffPool over the pathway graph (vae.py:238-243): the projection writes the batch of pathway graphs
        # [B * groups * k, 146, hidden] -- p.reshape(B, hidden, 146, groups * k).permute(0, 3, 2, 1).reshape(-1, 146,
        # hidden) of the reference's [B, C, 438, k] result -- directly
        z = segment_project(h, batch.gene_pca_match, batch.raw_indice, self.learnable_pca_params,
                            batch.nodes_per_graph, N_PATHWAYS * N_GROUPS, match_mask=True, pooled_groups=N_GROUPS)
        z, link, ent = self.diff_pooling(z, self.pathway_adj)
        return F.softmax(self.head(z.reshape(B, -1)), dim=-1), link, ent


def training_loss(model, batch, aux=True):
    """BCE on the class probabilities (train.py:118,60) + the DiffPool auxiliary losses.  ``aux=False``: BCE only -- the
    link loss is ONE Frobenius norm over the batch (``dense_diff_pool``), not a mean over graphs, so only the BCE part of
    a data-parallel step equals the single-process step on the global batch (tests/test_bench_gpu.py)."""
    pred, link, ent = model(batch)
    bce = F.binary_cross_entropy(pred, batch.y.reshape(-1, 2))
    return bce + link + ent if aux else bce
