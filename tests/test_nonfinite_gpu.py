"""Non-finite inputs.  The reference carries a NaN in a node feature or an edge attribute to the loss
(``relu(NaN) = NaN``, models/gcn_lib/sparse/torch_vertex.py:94-101; LayerNorm + ReLU, torch_nn.py:54-75); kernels
built on v_max / v_min would drop it silently and a diverged run would keep training.  The aggregation kernels carry an
explicit non-finite tracker (csrc/aggregate_common.h), LayerNorm / GEMM-prologue ReLUs keep NaN.  What is asserted:
the NaN pattern of the HIP result equals the CPU oracle's, finite entries still agree, +-Inf poisons too."""
from types import SimpleNamespace

import pytest
import torch

from _util import assert_close, golden_files, literal, load_golden, make_args
from oracle import gcn_lib as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(gen, N, E):
    src = torch.randint(0, N, (E,), generator=gen)
    dst = torch.randint(0, N - 2, (E,), generator=gen)
    return torch.stack([src, dst])


@pytest.mark.parametrize("aggr", ["softmax", "softmax_sg", "max", "mean", "add", "power"])
@pytest.mark.parametrize("where", ["x", "edge"])
@pytest.mark.parametrize("d", [32, 128])
def test_nan_reaches_every_destination(aggr, where, d):
    from mlgnn import CSRGraph, RankOneEdge, gen_aggregate
    gen = torch.Generator().manual_seed(17)
    N, E = 300, 2400
    ei = _graph(gen, N, E)
    x = torch.randn(N, d, generator=gen)
    a = torch.rand(E, generator=gen)
    u = torch.randn(d, generator=gen) * 0.5
    v = torch.randn(d, generator=gen) * 0.2
    if where == "x":
        x[7, 3] = float("nan")           # one channel of one source node
        x[11] = float("nan")             # a whole source row
    else:
        a[5] = float("nan")              # one edge: every channel of its message
    msg = torch.relu(x[ei[0]] + a[:, None] * u + v) + 1e-7
    ref = G.gen_aggregate(msg, ei[1], N, aggr, t=1.0, p=2.0)
    out = gen_aggregate(x.to(DEV), CSRGraph(ei.to(DEV), N), RankOneEdge(a.to(DEV), u.to(DEV), v.to(DEV)), aggr=aggr,
                        t=1.0, p=2.0).cpu()
    want_nan = torch.isnan(ref)
    # max: torch_scatter's comparison-based scatter_max never lets a NaN message win (oracle/primitives.py follows it)
    assert (want_nan.any() or aggr == "max") and not want_nan.all()
    assert torch.equal(torch.isnan(out), want_nan), "NaN pattern differs: %d vs %d entries" % (
        int(torch.isnan(out).sum()), int(want_nan.sum()))
    fin = ~want_nan
    assert_close(torch.where(fin, out, torch.zeros_like(out)), torch.where(fin, ref, torch.zeros_like(ref)), 1e-4,
                 "finite entries next to NaN rows", elementwise=True)


@pytest.mark.parametrize("aggr", ["softmax", "max", "mean"])
def test_inf_poisons_the_row(aggr):
    from mlgnn import CSRGraph, gen_aggregate
    gen = torch.Generator().manual_seed(18)
    N, E, d = 200, 1500, 64
    ei = _graph(gen, N, E)
    x = torch.randn(N, d, generator=gen)
    x[9, 5] = float("inf")
    out = gen_aggregate(x.to(DEV), CSRGraph(ei.to(DEV), N), None, aggr=aggr).cpu()
    hit = torch.zeros(N, dtype=torch.bool)
    hit[ei[1][ei[0] == 9]] = True
    assert hit.any()
    assert (~torch.isfinite(out[hit, 5])).all()
    assert torch.isfinite(out[~hit]).all()


def test_layernorm_relu_keeps_nan():
    from mlgnn.norm import layer_norm_act
    gen = torch.Generator().manual_seed(19)
    x = torch.randn(500, 128, generator=gen)
    x[17, 40] = float("nan")
    w, b = torch.rand(128, generator=gen) + 0.5, torch.randn(128, generator=gen)
    y = layer_norm_act(x.to(DEV), w.to(DEV), b.to(DEV), relu=True).cpu()
    ref = torch.relu(torch.nn.functional.layer_norm(x, (128,), w, b))
    assert torch.equal(torch.isnan(y), torch.isnan(ref)) and torch.isnan(y[17]).all()
    assert_close(torch.nan_to_num(y), torch.nan_to_num(ref), 1e-5, "finite rows")


@pytest.mark.parametrize("over", [dict(gcn_aggr="softmax"), dict(gcn_aggr="max", block="plain"),
                                  dict(gcn_aggr="mean", block="res")])
@pytest.mark.parametrize("where", ["x", "edge_attr"])
def test_model_output_is_nan_like_the_reference(over, where):
    """DeeperGCN on two graphs, a NaN planted in the first one: its prediction is NaN (loss NaN, as in the reference),
    the second graph's prediction is untouched."""
    from models import get_model
    from oracle import models as M
    f = load_golden([p for p in golden_files("deepergcn") if p.endswith("deepergcn_1.npz")][0])
    base = dict(num_layers=3, hidden_channels=32, dropout=0.0, conv_encode_edge=True, use_edge_attr=True, use_column="w",
                global_edge="none", graph_pooling="mean", norm="layer", mlp_layers=2, block="res+",
                pathway_global_node=False, node_embedding=False, use_age=False, num_layer_head=1, pathway_num=8,
                pathway_readout=None)
    args = make_args(**dict(base, **over))
    torch.manual_seed(3)
    model = get_model("deepergcn")(args)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = SimpleNamespace(**{k: f[k].clone() for k in ("x", "edge_index", "edge_attr", "batch", "age",
                                                         "pathway_node_attr", "node_size")})
    clean = M.deepergcn_forward(args, sd, batch, training=False)
    if where == "x":
        batch.x[3, 1] = float("nan")
    else:
        first = int((batch.batch[batch.edge_index[1]] == 0).nonzero()[0])
        batch.edge_attr[first] = float("nan")
    ref = M.deepergcn_forward(args, sd, batch, training=False)
    model.to(DEV).eval()
    out = model(SimpleNamespace(**{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in vars(batch).items()})).cpu()
    assert torch.isfinite(ref[1]).all()
    if over["gcn_aggr"] != "max" or where == "x":        # a NaN edge message is ignored by scatter_max, as in the reference
        assert torch.isnan(ref[0]).all()
    assert torch.equal(torch.isnan(out), torch.isnan(ref))
    assert_close(out[1], clean[1], 1e-4, "the other graph")
