#!/usr/bin/env python3
"""Generate the golden fixtures under ``tests/golden/`` from the reference's OWN classes.

Run here (authoring container) only:  ``python tests/golden/make_golden.py``.
The reference (``/root/reference``) never travels; the ``.npz`` files written
next to this script do.  Nothing in the test-suite imports this file.

How the reference is made importable: its hot-path modules import
``torch_geometric`` / ``torch_scatter`` / ``torch_cluster`` / ``h5py``, none of
which is installed or installable here (ordinary ``ModuleNotFoundError``; no
permission was denied).  This script registers *import-only* modules under
those names whose few entry points the hot path really calls are bound to
``oracle.primitives``.  Split of authority ("Oracle"):

* everything authored in the reference tree -- message formula, aggregator
  composition, MsgNorm, MLP layout, residual wiring, SAGE message/update,
  projection pooling, conv head, feature loss, DiffPool wiring, ``state_dict``
  keys -- is executed from the reference's source and therefore PINNED by the
  fixtures;
* the third-party primitives are our restatement: **parity unpinned**.
"""
import inspect
import math
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MLGNN_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
from oracle import primitives as P  # noqa: E402


# ----------------------------------------------------------------------------
# import-only stand-ins for the absent third-party modules
# ----------------------------------------------------------------------------
class _PygLinear(nn.Module):
    """torch_geometric.nn.dense.linear.Linear: NOT an nn.Linear subclass (so the
    reference's xavier ``init_weight`` skips it); kaiming-uniform(a=sqrt 5)."""

    def __init__(self, in_channels, out_channels, bias=True, **kw):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(in_channels)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return torch.nn.functional.linear(x, self.weight, self.bias)


class _MessagePassing(nn.Module):
    """Argument-by-name ``propagate`` of torch_geometric 2.2.0, flow source->target."""

    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2, **kw):
        super().__init__()
        self.aggr = aggr
        self.node_dim = node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        x = kwargs.get("x")
        n = x.size(0) if size is None else size[1]
        pool = dict(kwargs)
        for k, v in kwargs.items():
            if torch.is_tensor(v) and k != "edge_attr" and v.size(0) == x.size(0):
                pool[k + "_j"] = v.index_select(0, edge_index[0])
                pool[k + "_i"] = v.index_select(0, edge_index[1])
        margs = [p for p in inspect.signature(self.message).parameters]
        msg = self.message(**{k: pool.get(k) for k in margs})
        out = self.aggregate(msg, edge_index[1], ptr=None, dim_size=n)
        uargs = [p for p in inspect.signature(self.update).parameters][1:]
        return self.update(out, **{k: pool.get(k) for k in uargs})

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        return P.scatter(inputs, index, dim_size, {"add": "sum"}.get(self.aggr, self.aggr))

    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out


class _SAGEConvBase(_MessagePassing):
    """Constructor surface of torch_geometric.nn.SAGEConv 2.2.0 (aggr='mean', lin_l, lin_r)."""

    def __init__(self, in_channels, out_channels, normalize=False, root_weight=True, bias=True, **kw):
        super().__init__(aggr="mean")
        self.in_channels, self.out_channels, self.normalize = in_channels, out_channels, normalize
        self.lin_l = _PygLinear(in_channels, out_channels, bias=bias)
        self.lin_r = _PygLinear(in_channels, out_channels, bias=False)


class _DenseSAGEConv(nn.Module):
    def __init__(self, in_channels, out_channels, normalize=False, bias=True):
        super().__init__()
        self.normalize = normalize
        self.lin_rel = _PygLinear(in_channels, out_channels, bias=False)
        self.lin_root = _PygLinear(in_channels, out_channels, bias=bias)

    def forward(self, x, adj, mask=None):
        assert mask is None
        return P.dense_sage_conv(x, adj, self.lin_rel.weight, self.lin_root.weight, self.lin_root.bias,
                                 self.normalize)


def _install_import_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Any(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    mod("h5py")
    mod("torch_cluster", knn_graph=None)
    mod("torch_scatter",
        scatter=lambda src, index, dim=-1, out=None, dim_size=None, reduce="sum":
            P.scatter(src, index, dim_size, reduce),
        scatter_softmax=lambda src, index, dim=-1, dim_size=None:
            P.scatter_softmax(src, index, int(index.max()) + 1 if dim_size is None else dim_size),
        scatter_add=None, scatter_mean=None, scatter_max=None, scatter_min=None)
    tg_nn = mod("torch_geometric.nn", MessagePassing=_MessagePassing, SAGEConv=_SAGEConvBase,
                DenseSAGEConv=_DenseSAGEConv, DenseGraphConv=_Any,
                dense_diff_pool=lambda x, adj, s, mask=None: P.dense_diff_pool(x, adj, s),
                global_add_pool=lambda x, b: P.global_pool(x, b, "sum"),
                global_mean_pool=lambda x, b: P.global_pool(x, b, "mean"),
                global_max_pool=lambda x, b: P.global_pool(x, b, "max"),
                TopKPooling=_Any, EdgeConv=_Any, GATConv=_Any, GCNConv=_Any, GINConv=_Any,
                DynamicEdgeConv=_Any)
    tg_utils = mod("torch_geometric.utils", degree=lambda index, num_nodes=None, dtype=None:
                   P.degree(index, num_nodes), remove_self_loops=P.remove_self_loops,
                   add_self_loops=lambda ei, ea=None, fill_value=None, num_nodes=None:
                   P.add_self_loops(ei, ea, 1.0, num_nodes),
                   to_dense_batch=None, to_dense_adj=None, scatter_=None)
    tg_data = mod("torch_geometric.data", Data=object, InMemoryDataset=object, DataLoader=object,
                  Batch=object, extract_zip=None, download_url=None)
    mod("torch_geometric.nn.conv", MessagePassing=_MessagePassing)
    mod("torch_geometric", nn=tg_nn, utils=tg_utils, data=tg_data)


def _reference():
    _install_import_stubs()
    sys.path.insert(0, REF)
    import opt as ref_opt  # noqa
    from models.gcn_lib.sparse import torch_message, torch_vertex  # noqa
    from models import deepergcn, multilevel_gnn, multilevel_gnn_seq, diff_pooling, vae, vq_vae, autoencoder  # noqa
    return SimpleNamespace(opt=ref_opt, msg=torch_message, vertex=torch_vertex, deepergcn=deepergcn,
                           mlg=multilevel_gnn, mlgseq=multilevel_gnn_seq, diffpool=diff_pooling, vae=vae,
                           vqvae=vq_vae, ae=autoencoder)


# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------
def default_args(ref, **over):
    argv, sys.argv = sys.argv, [sys.argv[0]]
    try:
        a = ref.opt.parser.parse_args([])
    finally:
        sys.argv = argv
    for k, v in over.items():
        setattr(a, k, v)
    return a


def small_graph(gen, n_graphs=2, n=64, e=256, edge_dim=1, weights=False):
    """Directed multigraph batch with duplicate edges, self loops and two isolated nodes per graph."""
    srcs, dsts = [], []
    for g in range(n_graphs):
        s = torch.randint(0, n - 2, (e,), generator=gen)      # nodes n-2, n-1 stay isolated
        d = torch.randint(0, n - 2, (e,), generator=gen)
        s[:8] = d[:8]                                         # self loops
        s[8:16], d[8:16] = s[16:24], d[16:24]                 # duplicates
        srcs.append(s + g * n)
        dsts.append(d + g * n)
    ei = torch.stack([torch.cat(srcs), torch.cat(dsts)])
    if weights:
        ea = torch.rand(ei.shape[1], edge_dim, generator=gen) * 2 - 1      # incl. negative weights
    else:
        ea = torch.rand(ei.shape[1], edge_dim, generator=gen)
    batch = torch.arange(n_graphs).repeat_interleave(n)
    return ei, ea, batch


PACK_ABOVE = 200          # dicts with more entries than this (438 decoder blocks x 4 tensors) are stored packed


def save(name, **arrays):
    """One .npz per fixture.  A dict becomes ``key/sub`` entries; a large dict is packed into three arrays
    (``key/__names__``, ``key/__shapes__`` [n, 4] padded with -1, ``key/__flat__`` float32) -- a zip member per tiny
    tensor costs more than the tensor.  ``tests/_util.py::load_golden`` undoes both."""
    out = {}
    for k, v in arrays.items():
        if isinstance(v, dict) and len(v) > PACK_ABOVE:
            as_np = {n: (t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)) for n, t in v.items()}
            for n, a in as_np.items():                    # e.g. BatchNorm's int64 num_batches_tracked: stored as is
                if a.dtype != np.float32 or a.ndim > 4:
                    out["%s/%s" % (k, n)] = a
            names = sorted(n for n, a in as_np.items() if a.dtype == np.float32 and a.ndim <= 4)
            arrs = [as_np[n] for n in names]
            out[k + "/__names__"] = np.array(names)
            out[k + "/__shapes__"] = np.array([list(a.shape) + [-1] * (4 - a.ndim) for a in arrs], dtype=np.int64)
            out[k + "/__flat__"] = np.concatenate([a.reshape(-1) for a in arrs]) if arrs else np.zeros(0, np.float32)
        elif isinstance(v, dict):
            for kk, vv in v.items():
                out["%s/%s" % (k, kk)] = vv.detach().cpu().numpy() if torch.is_tensor(vv) else np.asarray(vv)
        else:
            out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %-40s %7.1f KB" % (os.path.basename(path), os.path.getsize(path) / 1024))


def grads_of(loss, named):
    names = [k for k, v in named.items() if v.requires_grad]
    gs = torch.autograd.grad(loss, [named[k] for k in names], allow_unused=True)
    return {k: (g if g is not None else torch.zeros_like(named[k])) for k, g in zip(names, gs)}


def probe_weights(out, gen):
    """Fixed random cotangent so that ``loss = sum(out * c)`` exercises every output element."""
    return torch.randn(out.shape, generator=gen)


# ----------------------------------------------------------------------------
# fixtures
# ----------------------------------------------------------------------------
def fx_aggregators(ref):
    gen = torch.Generator().manual_seed(101)
    N, E, d = 64, 256, 16
    index = torch.randint(0, N - 2, (E,), generator=gen)
    base = torch.rand(E, d, generator=gen) * 3 + 1e-7
    base[::7] = 1e-7                      # ties at the relu floor
    base[5] = 12.0                        # above the power clamp
    cases = [("add", {}), ("mean", {}), ("max", {}),
             ("softmax", dict(t=1.0)), ("softmax", dict(t=0.5, learn_t=True)),
             ("softmax_sg", dict(t=2.0)), ("softmax_sum", dict(t=1.0, learn_t=True, y=0.3, learn_y=True)),
             ("power", dict(p=2.0)), ("power", dict(p=3.0, learn_p=True)),
             ("power_sum", dict(p=2.0, y=-0.2, learn_y=True))]
    blob = dict(index=index, inputs=base, n_nodes=N)
    for ci, (aggr, kw) in enumerate(cases):
        mp = ref.msg.GenMessagePassing(aggr=aggr, **kw)
        inp = base.clone().requires_grad_(True)
        out = mp.aggregate(inp * 1.0, index, dim_size=N)      # *1.0: power clamps its input in place
        c = probe_weights(out, gen)
        named = {"inputs": inp}
        named.update({k: v for k, v in mp.named_parameters()})
        g = grads_of((out * c).sum(), named)
        tag = "c%d" % ci
        blob[tag + "/aggr"] = np.array(aggr)
        blob[tag + "/kw"] = np.array(repr(sorted(kw.items())))
        blob[tag + "/out"] = out
        blob[tag + "/cot"] = c
        for k, v in g.items():
            blob[tag + "/grad/" + k] = v
    blob["n_cases"] = len(cases)
    save("aggregators", **blob)


def fx_genconv(ref):
    gen = torch.Generator().manual_seed(202)
    cfgs = [dict(d=32, aggr="softmax", t=1.0, learn_t=False, msg_norm=False, norm="layer"),
            dict(d=16, aggr="softmax", t=0.7, learn_t=True, msg_norm=True, learn_msg_scale=True, norm="layer"),
            dict(d=32, aggr="max", norm="layer"),
            dict(d=16, aggr="mean", norm="batch"),
            dict(d=16, aggr="power", p=2.0, norm="layer"),
            dict(d=32, aggr="softmax_sg", t=1.5, norm="layer", edge_dim_full=True),
            # degree-scaled aggregators (torch_message.py:60-63,77-80), y learnable / fixed (appended: the
            # generator state of the fixtures above is unchanged)
            dict(d=16, aggr="softmax_sum", t=0.8, learn_t=True, y=0.3, learn_y=True, norm="layer"),
            dict(d=32, aggr="power_sum", p=2.0, y=-0.4, learn_y=False, norm="layer"),
            dict(d=16, aggr="power_sum", p=1.5, learn_p=True, y=0.2, learn_y=True, msg_norm=True, norm="layer")]
    for ci, cfg in enumerate(cfgs):
        cfg = dict(cfg)
        d = cfg.pop("d")
        full = cfg.pop("edge_dim_full", False)
        ei, ea, _ = small_graph(gen, 2, 64, 256)
        N = 128
        torch.manual_seed(300 + ci)
        conv = ref.vertex.GENConv(d, d, encode_edge=True, edge_feat_dim=(d if full else 1), mlp_layers=2, **cfg)
        conv.train()
        x = torch.randn(N, d, generator=gen, requires_grad=True)
        if full:
            ea = torch.randn(ei.shape[1], d, generator=gen)
        ea = ea.clone().requires_grad_(True)
        out = conv(x, ei, ea)
        c = probe_weights(out, gen)
        named = {"x": x, "edge_attr": ea}
        named.update({"sd." + k: v for k, v in conv.named_parameters()})
        g = grads_of((out * c).sum(), named)
        save("genconv_%d" % ci, cfg=np.array(repr(sorted(dict(cfg, d=d, full=full).items()))),
             x=x, edge_index=ei, edge_attr=ea, out=out, cot=c, sd=dict(conv.state_dict()), grad=g)


def fx_sage(ref):
    gen = torch.Generator().manual_seed(303)
    for ci, (kind, cin, cout) in enumerate([("sage", 16, 32), ("rsage", 32, 16), ("sage", 1, 8)]):
        ei, ea, _ = small_graph(gen, 2, 64, 256, weights=True)
        torch.manual_seed(400 + ci)
        conv = ref.vertex.GraphConv(cin, cout, conv=kind, act="leakyrelu", mlp_norm="none")
        x = torch.randn(128, cin, generator=gen, requires_grad=True)
        out = conv(x, ei, ea)
        c = probe_weights(out, gen)
        named = {"x": x}
        named.update({"sd." + k: v for k, v in conv.named_parameters()})
        g = grads_of((out * c).sum(), named)
        save("sage_%d" % ci, kind=np.array(kind), x=x, edge_index=ei, edge_attr=ea, out=out, cot=c,
             sd=dict(conv.state_dict()), grad=g)


def fx_deepergcn(ref):
    gen = torch.Generator().manual_seed(404)
    base = dict(num_layers=3, hidden_channels=32, dropout=0.0, conv_encode_edge=True, use_edge_attr=True,
                use_column="w", global_edge="none", graph_pooling="mean", norm="layer", mlp_layers=2,
                block="res+", pathway_global_node=False, node_embedding=False, use_age=False,
                num_layer_head=1, pathway_num=8, pathway_readout=None)
    cases = [dict(gcn_aggr="max"), dict(gcn_aggr="softmax"), dict(gcn_aggr="softmax_sg", t=2.0),
             dict(gcn_aggr="mean"), dict(gcn_aggr="add"), dict(gcn_aggr="power", p=2.0),
             dict(gcn_aggr="softmax", learn_t=True, t=0.5, msg_norm=True, learn_msg_scale=True),
             dict(gcn_aggr="softmax", block="res", graph_pooling="max"),
             dict(gcn_aggr="max", block="plain", graph_pooling="sum", num_layer_head=2, use_age=True),
             dict(gcn_aggr="softmax", pathway_global_node=True, pathway_readout="maxpool", pre_concat_age=True,
                  pre_readout_drop=True, use_age=True, num_layer_head=2),
             dict(gcn_aggr="softmax", global_edge="onehot", pathway_edge_num=5),
             # use_column None: the 7-column edge attribute through Linear(7, hidden) (deepergcn.py:90)
             dict(gcn_aggr="softmax", use_column=None),
             dict(gcn_aggr="max", use_column=None, block="res", msg_norm=True, learn_msg_scale=True),
             # degree-scaled aggregators through the whole model (appended)
             dict(gcn_aggr="softmax_sum", learn_t=True, t=0.7, y=0.25, learn_y=True),
             dict(gcn_aggr="power_sum", p=2.0, y=-0.3, learn_y=True, block="res")]
    for ci, over in enumerate(cases):
        a = default_args(ref, **dict(base, **over))
        ei, ea, bvec = small_graph(gen, 2, 64, 256, edge_dim=7 if a.use_column is None else 1)
        if a.global_edge == "onehot":
            ea = torch.randint(0, 5, (ei.shape[1], 1), generator=gen).to(torch.float32)
        torch.manual_seed(500 + ci)
        model = ref.deepergcn.DeeperGCN(a)
        model.train()                       # dropout p=0; LayerNorm: train == eval
        x = torch.randn(128, 3, generator=gen)
        batch = SimpleNamespace(x=x, edge_index=ei, edge_attr=ea, batch=bvec,
                                age=torch.rand(2, generator=gen),
                                pathway_node_attr=torch.randn(2 * 8, 6, generator=gen),   # [B*pn, 6]: the only layout :223 accepts
                                node_size=torch.tensor([64, 64]))
        out = model(batch)
        c = probe_weights(out, gen)
        named = {"sd." + k: v for k, v in model.named_parameters()}
        g = grads_of((out * c).sum(), named)
        save("deepergcn_%d" % ci, over=np.array(repr(sorted(over.items()))), x=x, edge_index=ei, edge_attr=ea,
             batch=bvec, age=batch.age, pathway_node_attr=batch.pathway_node_attr, node_size=batch.node_size,
             out=out, cot=c, sd=dict(model.state_dict()), grad=g)


def fx_multilevel(ref):
    gen = torch.Generator().manual_seed(505)
    node_num, B, G, S = 40, 2, 900, 438
    for ci, over in enumerate([dict(gnn_name="sage"), dict(gnn_name="rsage", resgnn=False, pca_dim=3, pca_pool_dim=1,
                                                          pathway_pool_dim=1, use_age=False, node_embedding_dim=32),
                               # GNN-loop wirings of multilevel_gnn.py:188-199 (appended): dense concatenation,
                               # residual layers, the input mask re-applied between layers (+ row normalisation)
                               dict(gnn_name="sage", dense_gnn=True, num_layers=3),
                               dict(gnn_name="sage", resgnn=True, num_layers=3, hidden_channels=16, final_channels=16,
                                    final_head=1),
                               dict(gnn_name="rsage", repeat_mask=True, repeat_cyclic=1, num_layers=3),
                               dict(gnn_name="sage", repeat_mask=True, repeat_cyclic=2, repeat_norm=True, num_layers=4,
                                    resgnn=True, hidden_channels=16, final_channels=16, final_head=1)]):
        kw = dict(model="multilevel_gnn", num_layers=2, hidden_channels=16, final_channels=8, final_head=4,
                  node_embedding=True, node_embedding_dim=16, gnn_name="sage", head_dim=4, use_age=True,
                  weighted_edge=True, value_att_mask=True, pca_match_mask=True, mutual_info_mask=True,
                  learnable_pca=True, pca_indep_loss=True, pca_loss=True, feature_drop=False, dropout=0.0,
                  conv_channel_list=[8, 8], conv_kernel_list=[1, 1])
        kw.update(over)
        a = default_args(ref, **kw)
        torch.manual_seed(600 + ci)
        model = ref.mlg.MultilevelGNN(a)
        NN = node_num * 3
        model.node_num = node_num
        model.node_embedding = nn.Parameter(torch.randn(NN, a.node_embedding_dim, generator=gen) * 0.3)
        mask = (torch.rand(G, generator=gen) > 0.2).to(torch.float32)
        comps = torch.randn(int(mask.sum()), a.pca_dim + 1, generator=gen) * 0.2
        model.set_pca_params(comps, mask)
        model.set_info_mask(mask[:, None].clone())
        seg = torch.sort(torch.randint(0, S, (G,), generator=gen))[0]
        model.set_pathway_indexs(seg.clone())
        model.eval()          # Dropout(0.5) in the head is p>0: fixtures are taken in eval mode
        ei, ea, _ = small_graph(gen, B, NN, 400, weights=True)
        match = torch.randint(0, NN, (B, G), generator=gen)
        match[:, ::11] = -1
        batch = SimpleNamespace(x=torch.rand(B * NN, 1, generator=gen), edge_index=ei, edge_attr=ea,
                                gene_pca_match=match, raw_indice=seg[None, :].repeat(B, 1),
                                age=torch.rand(B, generator=gen))
        pred, feat = model(batch)
        floss = model.get_feature_loss(feat)
        c = probe_weights(pred, gen)
        named = {"sd." + k: v for k, v in model.named_parameters()}
        loss = (pred * c).sum() + floss
        g = grads_of(loss, named)
        save("multilevel_%d" % ci, over=np.array(repr(sorted(kw.items()))), node_num=node_num, x=batch.x,
             edge_index=ei, edge_attr=ea, gene_pca_match=match, raw_indice=batch.raw_indice, age=batch.age,
             pathway_indexs=seg, pred=pred, pca_feature=feat, feature_loss=floss, cot=c,
             sd=dict(model.state_dict()), grad=g)


def fx_mlgseq(ref):
    """``MultilevelGNNSeq`` (models/multilevel_gnn_seq.py): the same GNN + projection pooling with the conv head
    factored into ``PathwayHeadSeq`` (state_dict keys ``pathwayhead.*``) and the ``only_mrna_pred`` column cut."""
    gen = torch.Generator().manual_seed(808)
    node_num, B, G, S = 40, 2, 900, 438
    for ci, over in enumerate([dict(), dict(only_mrna_pred=True, use_age=False, gnn_name="rsage", resgnn=False,
                                            pca_pool_dim=1, pathway_pool_dim=2)]):
        kw = dict(model="multilevel_gnn_seq", num_layers=2, hidden_channels=16, final_channels=8, final_head=4,
                  node_embedding=True, node_embedding_dim=16, gnn_name="sage", head_dim=4, use_age=True,
                  weighted_edge=True, value_att_mask=True, pca_match_mask=True, mutual_info_mask=True,
                  learnable_pca=True, pca_indep_loss=True, pca_loss=True, feature_drop=False, dropout=0.0,
                  conv_channel_list=[8, 8], conv_kernel_list=[1, 1])
        kw.update(over)
        a = default_args(ref, **kw)
        torch.manual_seed(900 + ci)
        model = ref.mlgseq.MultilevelGNNSeq(a)
        NN = node_num * 3
        model.node_num = node_num
        model.node_embedding = nn.Parameter(torch.randn(NN, a.node_embedding_dim, generator=gen) * 0.3)
        mask = (torch.rand(G, generator=gen) > 0.2).to(torch.float32)
        comps = torch.randn(int(mask.sum()), a.pca_dim + 1, generator=gen) * 0.2
        model.set_pca_params(comps, mask)
        model.set_info_mask(mask[:, None].clone())
        seg = torch.sort(torch.randint(0, S, (G,), generator=gen))[0]
        model.set_pathway_indexs(seg.clone())
        model.eval()
        ei, ea, _ = small_graph(gen, B, NN, 400, weights=True)
        match = torch.randint(0, NN, (B, G), generator=gen)
        match[:, ::11] = -1
        batch = SimpleNamespace(x=torch.rand(B * NN, 1, generator=gen), edge_index=ei, edge_attr=ea,
                                gene_pca_match=match, raw_indice=seg[None, :].repeat(B, 1),
                                age=torch.rand(B, generator=gen))
        pred, feat = model(batch)
        floss = model.get_feature_loss(feat)
        c = probe_weights(pred, gen)
        named = {"sd." + k: v for k, v in model.named_parameters()}
        loss = (pred * c).sum() + floss
        g = grads_of(loss, named)
        save("mlgseq_%d" % ci, over=np.array(repr(sorted(kw.items()))), node_num=node_num, x=batch.x,
             edge_index=ei, edge_attr=ea, gene_pca_match=match, raw_indice=batch.raw_indice, age=batch.age,
             pathway_indexs=seg, pred=pred, pca_feature=feat, feature_loss=floss, cot=c,
             sd=dict(model.state_dict()), grad=g)


def fx_vae(ref):
    """``VAE`` (models/vae.py): encoder (GNN + projection pooling + mu / sigma heads and its losses), the per-pathway
    decoders, ``train_step`` -> ``predict_head`` with DiffPool on the pathway graph ('pathway' and 'head' placement) or
    the conv + max-pool head, ``reconstruct_head`` sizing, and the deterministic parts of ``vae_loss``."""
    gen = torch.Generator().manual_seed(909)
    node_num, B, G, S = 40, 3, 900, 438
    cases = [dict(reorder_type="diff_pooling", diff_pooling_location="pathway"),
             dict(reorder_type="diff_pooling", diff_pooling_location="head", gnn_name="rsage", resgnn=False,
                  after_pooling_layer=2, use_age=False, channel_one=True, final_channels=1,
                  final_head=1),
             dict(reorder_type="pca", channel_one=True, final_channels=1, final_head=1, pathway_pool_dim=2,
                  decoder_type="foreach_diffhidden")]
    for ci, over in enumerate(cases):
        kw = dict(model="vae", num_layers=2, hidden_channels=16, final_channels=4, final_head=2,
                  node_embedding=True, node_embedding_dim=16, gnn_name="sage", head_dim=8, use_age=True,
                  weighted_edge=True, pca_match_mask=True, mutual_info_mask=True, pca_dim=2,
                  feature_drop=False, dropout=0.0, conv_channel_list=[8, 8], conv_kernel_list=[1, 1],
                  decoder_type="foreach", decoder_dim=4, diff_pooling_layer=2, diff_pooling_hidden_dim=8,
                  diff_pooling_output_dim=6, pathway_num=146, after_pooling_layer=1)
        kw.update(over)
        a = default_args(ref, **kw)
        seg = torch.randint(0, S, (G,), generator=gen)
        seg[:S] = torch.arange(S)                               # every pathway x omics segment owns a decoder
        seg = torch.sort(seg)[0]
        torch.manual_seed(950 + ci)
        model = ref.vae.VAE(a, None, seg)
        NN = node_num * 3
        model.node_num = node_num
        model.node_embedding = nn.Parameter(torch.randn(NN, a.node_embedding_dim, generator=gen) * 0.3)
        mask = (torch.rand(G, generator=gen) > 0.2).to(torch.float32)
        first = torch.searchsorted(seg, torch.arange(S))        # one live, matched gene per segment: a segment without
        mask[first] = 1.0                                       # any gives a constant mu row and a NaN corrcoef
        model.set_pca_params(torch.randn(int(mask.sum()), a.pca_dim + 1, generator=gen) * 0.2, mask)
        model.set_info_mask(mask[:, None].clone())
        model.set_pathway_indexs(seg.clone())
        sim = np.abs(np.corrcoef(np.random.RandomState(ci).randn(146, 20))) - np.eye(146)
        model.set_pathway_similarity_matrix(sim)
        model.reconstruct_head(a)
        model.eval()
        ei, ea, _ = small_graph(gen, B, NN, 400, weights=True)
        match = torch.randint(0, NN, (B, G), generator=gen)
        match[:, ::11] = -1
        match[:, first] = torch.randint(0, NN, (B, S), generator=gen)
        batch = SimpleNamespace(x=torch.rand(B * NN, 1, generator=gen), edge_index=ei, edge_attr=ea,
                                gene_pca_match=match, raw_indice=seg[None, :].repeat(B, 1),
                                age=torch.rand(B, generator=gen))
        named = {"sd." + k: v for k, v in model.named_parameters()}
        # --- prediction path: train_step -> predict_head
        pred, feat, l, e, gene_feature = model.train_step(batch)
        c = probe_weights(pred, gen)
        l_t, e_t = torch.as_tensor(l, dtype=torch.float32), torch.as_tensor(e, dtype=torch.float32)
        g_pred = grads_of((pred * c).sum() + 0.7 * l_t + 0.3 * e_t, named)
        # --- reconstruction path with a given latent sample (rsample() itself is a random draw)
        q_z, h, enc_losses, _ = model.encoder(batch)
        z = (q_z.loc + 0.5 * q_z.scale).detach()
        recon = model.foreach_decoder(q_z.loc + 0.5 * q_z.scale)
        target = torch.rand(recon.shape, generator=gen)
        kld = torch.distributions.kl_divergence(q_z, torch.distributions.Normal(0, 1.)).sum(-1).mean()
        rec = F.mse_loss(recon, target)
        g_rec = grads_of(rec + 0.1 * kld + enc_losses[0] + enc_losses[2], named)
        save("vae_%d" % ci, over=np.array(repr(sorted(kw.items()))), node_num=node_num, x=batch.x,
             edge_index=ei, edge_attr=ea, gene_pca_match=match, raw_indice=batch.raw_indice, age=batch.age,
             pathway_indexs=seg, similarity=sim.astype(np.float32), pred=pred, pca_feature=feat, link=l_t, ent=e_t,
             gene_feature=gene_feature, cot=c, embedding=h, loss_std=enc_losses[0], loss_corr=enc_losses[2],
             z=z, recon=recon, target=target, kld=kld, rec=rec,
             sd=dict(model.state_dict()), grad_pred=g_pred, grad_rec=g_rec)


def _pretrain_model(ref, cls, kw, ci, gen, seed):
    """Shared set-up of the pre-training models' fixtures (VQ_VAE, AutoEncoder): small node_num, masked projection with
    one live matched gene per pathway segment, a 3-graph batch."""
    node_num, B, G, S = 40, 3, 900, 438
    a = default_args(ref, **kw)
    seg = torch.randint(0, S, (G,), generator=gen)
    seg[:S] = torch.arange(S)
    seg = torch.sort(seg)[0]
    torch.manual_seed(seed + ci)
    model = cls(a, None, seg)
    NN = node_num * 3
    model.node_num = node_num
    model.node_embedding = nn.Parameter(torch.randn(NN, a.node_embedding_dim, generator=gen) * 0.3)
    mask = (torch.rand(G, generator=gen) > 0.2).to(torch.float32)
    first = torch.searchsorted(seg, torch.arange(S))
    mask[first] = 1.0
    model.set_pca_params(torch.randn(int(mask.sum()), a.pca_dim + 1, generator=gen) * 0.2, mask)
    model.set_info_mask(mask[:, None].clone())
    model.set_pathway_indexs(seg.clone())
    model.eval()
    ei, ea, _ = small_graph(gen, B, NN, 400, weights=True)
    match = torch.randint(0, NN, (B, G), generator=gen)
    match[:, ::11] = -1
    match[:, 5::13] = 0                                       # node 0: kept by the VAEs, masked by the AutoEncoder
    match[:, first] = torch.randint(1, NN, (B, S), generator=gen)
    batch = SimpleNamespace(x=torch.rand(B * NN, 1, generator=gen), edge_index=ei, edge_attr=ea,
                            gene_pca_match=match, raw_indice=seg[None, :].repeat(B, 1),
                            age=torch.rand(B, generator=gen))
    common = dict(over=np.array(repr(sorted(kw.items()))), node_num=node_num, x=batch.x, edge_index=ei, edge_attr=ea,
                  gene_pca_match=match, raw_indice=batch.raw_indice, age=batch.age, pathway_indexs=seg)
    return a, model, batch, common


PRETRAIN_KW = dict(num_layers=2, hidden_channels=16, final_channels=4, final_head=2, node_embedding=True,
                   node_embedding_dim=16, gnn_name="sage", head_dim=8, use_age=True, weighted_edge=True,
                   pca_match_mask=True, mutual_info_mask=True, pca_dim=2, feature_drop=False, dropout=0.0,
                   conv_channel_list=[8, 8], conv_kernel_list=[1, 1], decoder_type="foreach", decoder_dim=4,
                   diff_pooling_layer=2, diff_pooling_hidden_dim=8, diff_pooling_output_dim=6, pathway_num=146,
                   after_pooling_layer=1)


def fx_vqvae(ref):
    """``VQ_VAE`` (models/vq_vae.py): encoder, ``VectorQuantizer``, decoders, ``train_step`` -> ``predict_head``.
    ``vqvae_beta`` is read by the constructor but has no ``opt.py`` entry: the fixture supplies 0.25 (the quantiser's
    own default)."""
    gen = torch.Generator().manual_seed(1010)
    for ci, over in enumerate([dict(reorder_type="diff_pooling", diff_pooling_location="pathway"),
                               dict(reorder_type="pca", channel_one=True, final_channels=1, final_head=1,
                                    gnn_name="rsage", resgnn=False, use_age=False)]):
        kw = dict(PRETRAIN_KW, model="vq_vae", vqvae_num_embeddings=32, vqvae_beta=0.25)
        kw.update(over)
        a, model, batch, common = _pretrain_model(ref, ref.vqvae.VQ_VAE, kw, ci, gen, 1050)
        sim = np.abs(np.corrcoef(np.random.RandomState(10 + ci).randn(146, 20))) - np.eye(146)
        model.set_pathway_similarity_matrix(sim)
        model.reconstruct_head(a)
        model.eval()
        with torch.no_grad():                                  # code words near the latents: several get selected
            zs = model.encoder(batch).reshape(-1, model.vq_layer.D)
            model.vq_layer.embedding.weight.copy_(zs[torch.randperm(zs.shape[0], generator=gen)[:32]] * 1.05)
        named = {"sd." + k: v for k, v in model.named_parameters()}
        pred, feat, l, e = model.train_step(batch)
        c = probe_weights(pred, gen)
        l_t, e_t = torch.as_tensor(l, dtype=torch.float32), torch.as_tensor(e, dtype=torch.float32)
        g_pred = grads_of((pred * c).sum() + 0.7 * l_t + 0.3 * e_t, named)
        out = model(batch)
        target = torch.rand(out["pred_x"].shape, generator=gen)
        terms = model.vae_loss(out["pred_x"], target, out["vq_loss"])
        g_rec = grads_of(terms["loss"], named)
        save("vqvae_%d" % ci, similarity=sim.astype(np.float32), pred=pred, pca_feature=feat, link=l_t, ent=e_t, cot=c,
             z=out["z"], quantized=out["embedding"], vq_loss=out["vq_loss"], recon=out["pred_x"], target=target,
             loss=terms["loss"], sd=dict(model.state_dict()), grad_pred=g_pred, grad_rec=g_rec, **common)


def fx_autoencoder(ref):
    """``AutoEncoder`` (models/autoencoder.py): encoder (masks ``match <= 0``) + per-pathway / flatten decoders."""
    gen = torch.Generator().manual_seed(1111)
    for ci, over in enumerate([dict(), dict(decoder_type="flatten", gnn_name="rsage", resgnn=False, final_channels=1,
                                            final_head=1)]):
        kw = dict(PRETRAIN_KW, model="autoencoder")
        kw.update(over)
        a, model, batch, common = _pretrain_model(ref, ref.ae.AutoEncoder, kw, ci, gen, 1150)
        named = {"sd." + k: v for k, v in model.named_parameters()}
        recon, h, _ = model(batch)
        c = probe_weights(recon, gen)
        g = grads_of((recon * c).sum(), named)
        save("autoencoder_%d" % ci, recon=recon, latent=h, cot=c, sd=dict(model.state_dict()), grad=g, **common)


def fx_diffpool(ref):
    gen = torch.Generator().manual_seed(606)
    for ci, (Bp, C, hid, outc, nl, apl) in enumerate([(4, 8, 32, 64, 2, 1), (3, 16, 16, 16, 1, 2)]):
        a = SimpleNamespace(pooling_type="correlation", after_pooling_layer=apl)
        torch.manual_seed(700 + ci)
        dp = ref.diffpool.DiffPool(C, None, 146, nl, hid, outc, a)
        dp.eval()
        x = torch.randn(Bp, 146, C, generator=gen, requires_grad=True)
        adj = torch.rand(146, 146, generator=gen)
        adj = (adj + adj.t()) / 2 + torch.eye(146)
        out, link, ent = dp(x, adj)
        c = probe_weights(out, gen)
        named = {"x": x}
        named.update({"sd." + k: v for k, v in dp.named_parameters()})
        g = grads_of((out * c).sum() + 0.7 * link + 0.3 * ent, named)
        save("diffpool_%d" % ci, cfg=np.array([Bp, C, hid, outc, nl, apl]), x=x, adj=adj, out=out, link=link,
             ent=ent, cot=c, sd=dict(dp.state_dict()), grad=g)


def main():
    ref = _reference()
    torch.set_num_threads(4)
    only = set(sys.argv[1:])               # e.g. `make_golden.py deepergcn` regenerates one family
    for name, fx in [("aggregators", fx_aggregators), ("genconv", fx_genconv), ("sage", fx_sage),
                     ("deepergcn", fx_deepergcn), ("multilevel", fx_multilevel), ("mlgseq", fx_mlgseq), ("vae", fx_vae), ("vqvae", fx_vqvae), ("autoencoder", fx_autoencoder),
                     ("diffpool", fx_diffpool)]:
        if not only or name in only:
            fx(ref)


if __name__ == "__main__":
    main()
