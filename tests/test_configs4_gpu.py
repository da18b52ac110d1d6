"""BASELINE configs[4] (one graph of 200 000 nodes / 3 000 000 edges, d = 256, 28-layer DeeperGCN, bf16 storage with
fp32 accumulation) as a tested configuration.

* kernels of a layer against the fp32 CPU oracle (not against the fp32 HIP path) on a graph whose feature table
  (48 000 x 256 bf16 = 24.6 MB) is past every XCD's 4 MB L2 -- the regime the stress run lives in;
* at the full size, where the oracle's [E, d] tensors would take minutes: the size-independent properties used for
  configs[1] (tests/test_workload_gpu.py::test_full_size_properties), a 4-layer training step, and one 28-layer
  forward for finiteness.
The 4096-node DiffPool of the same config is tests/test_diffpool_large_gpu.py.  Tolerances are bf16-sized and say so."""
from types import SimpleNamespace

import os

import pytest
import torch

from _util import assert_close, make_args
from oracle import gcn_lib as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16 = 2.0 ** -8            # one bf16 rounding (8 significant bits)


def _rb(t):
    return t.to(torch.bfloat16).float()


def test_bf16_layer_kernels_vs_fp32_cpu_oracle_past_l2():
    from mlgnn import CSRGraph, RankOneEdge, gen_aggregate
    from mlgnn.dense import linear
    from mlgnn.norm import layer_norm_act
    gen = torch.Generator().manual_seed(4)
    N, E, d = 48000, 700000, 256
    ei = torch.stack([torch.randint(0, N, (E,), generator=gen), torch.randint(0, N - 5, (E,), generator=gen)])
    x = _rb(torch.randn(N, d, generator=gen))
    a = torch.rand(E, generator=gen)
    u, v = torch.randn(d, generator=gen) * 0.5, torch.randn(d, generator=gen) * 0.2
    graph = CSRGraph(ei.to(DEV), N)
    xg = x.to(DEV).to(torch.bfloat16)
    # ---- aggregation (softmax / max / mean), fp32 oracle on the same bf16-rounded inputs -------------------------
    msg = torch.relu(x[ei[0]] + a[:, None] * u + v) + 1e-7
    for aggr in ("softmax", "max", "mean"):
        ref = G.gen_aggregate(msg, ei[1], N, aggr, t=1.0)
        out = gen_aggregate(xg, graph, RankOneEdge(a.to(DEV), u.to(DEV), v.to(DEV)), aggr=aggr, t=1.0)
        assert out.dtype == torch.bfloat16
        # fp32 arithmetic inside, ONE rounding at the store: elementwise 2^-8 relative (+ a floor for entries near 0)
        assert_close(out.float(), ref, BF16, "bf16 %s vs fp32 CPU oracle" % aggr, elementwise=True)
    del msg
    # ---- LayerNorm + ReLU -----------------------------------------------------------------------------------------
    w, b = torch.rand(d, generator=gen) + 0.5, torch.randn(d, generator=gen) * 0.3
    ref = torch.relu(torch.nn.functional.layer_norm(x, (d,), w, b))
    y = layer_norm_act(xg, w.to(DEV), b.to(DEV), relu=True)
    assert y.dtype == torch.bfloat16
    assert_close(y.float(), ref, BF16, "bf16 LayerNorm+ReLU vs fp32 CPU", elementwise=True)
    # ---- Linear 256 -> 512 (+ bias) and back 512 -> 256 with the residual, fp32 accumulation --------------------
    W1, b1 = _rb(torch.randn(512, d, generator=gen) / 16), torch.randn(512, generator=gen) * 0.1
    W2 = _rb(torch.randn(d, 512, generator=gen) / 22)
    h_ref = torch.nn.functional.linear(x, W1, b1)
    h = linear(xg, W1.to(DEV).to(torch.bfloat16), b1.to(DEV).to(torch.bfloat16))
    # bias is rounded to bf16 by the caller's dtype; one rounding of it and one of the result
    assert_close(h.float(), torch.nn.functional.linear(x, W1, _rb(b1)), BF16, "bf16 Linear vs fp32 CPU", elementwise=True)
    hr = _rb(h_ref)
    o_ref = torch.nn.functional.linear(hr, W2) + x
    o = linear(hr.to(DEV).to(torch.bfloat16), W2.to(DEV).to(torch.bfloat16), None, residual=xg)
    assert_close(o.float(), o_ref, BF16, "bf16 Linear + residual vs fp32 CPU", elementwise=True)


def _full_graph(gen):
    N, E = 200000, 3000000
    ei = torch.randint(0, N, (2, E), generator=gen)
    return N, E, ei


def test_configs4_full_size_aggregation_properties():
    from mlgnn import CSRGraph, RankOneEdge, gen_aggregate
    gen = torch.Generator().manual_seed(7)
    N, E, ei = _full_graph(gen)
    d = 256
    x = torch.randn(N, d, generator=gen).to(DEV).to(torch.bfloat16)
    a = torch.rand(E, generator=gen)
    u, v = (torch.randn(d, generator=gen) * 0.5).to(DEV), (torch.randn(d, generator=gen) * 0.2).to(DEV)
    perm = torch.randperm(E, generator=gen)
    g, g2 = CSRGraph(ei.to(DEV), N), CSRGraph(ei[:, perm].to(DEV), N)
    e1, e2 = RankOneEdge(a.to(DEV), u, v), RankOneEdge(a[perm].to(DEV), u, v)
    add = gen_aggregate(x, g, e1, aggr="add").float()
    # edge order only changes the fp32 summation order in front of one bf16 rounding
    assert_close(gen_aggregate(x, g2, e2, aggr="add").float(), add, BF16, "edge-order invariance (add)")
    mx = gen_aggregate(x, g, e1, aggr="max")
    assert torch.equal(gen_aggregate(x, g2, e2, aggr="max"), mx), "edge-order invariance (max) is exact"
    sm = gen_aggregate(x, g, e1, aggr="softmax").float()
    assert_close(gen_aggregate(x, g2, e2, aggr="softmax").float(), sm, BF16, "edge-order invariance (softmax)")
    mean = gen_aggregate(x, g, e1, aggr="mean").float()
    assert_close(mean * g.in_degree.clamp(min=1)[:, None], add, 2 * BF16, "mean * deg = add")
    has = (g.in_degree > 0)[:, None]
    assert bool(((sm <= mx.float() * (1 + BF16) + 1e-6) | ~has).all()) and bool(((sm >= 0) | ~has).all())
    assert bool((add[~has.squeeze(1)] == 0).all())


def _model(layers):
    from models import get_model
    args = make_args(num_layers=layers, hidden_channels=256, dropout=0.0, conv_encode_edge=True, use_edge_attr=True,
                     use_column="w", global_edge="none", gcn_aggr="softmax", block="res+", norm="layer",
                     graph_pooling="mean", pathway_readout=None)
    torch.manual_seed(0)
    return get_model("deepergcn")(args).to(DEV).to(torch.bfloat16)


def _batch(gen):
    N, E, ei = _full_graph(gen)
    return SimpleNamespace(x=torch.randn(N, 3, generator=gen).to(DEV).to(torch.bfloat16), edge_index=ei.to(DEV),
                           edge_attr=torch.rand(E, 1, generator=gen).to(DEV),
                           batch=torch.zeros(N, dtype=torch.long, device=DEV), age=torch.zeros(1, device=DEV, dtype=torch.bfloat16),
                           pathway_node_attr=None, node_size=torch.tensor([N], device=DEV))


def test_configs4_four_layer_training_step_and_28_layer_forward():
    gen = torch.Generator().manual_seed(7)
    batch = _batch(gen)
    model = _model(4)
    out = model(batch)
    assert out.shape == (1, 2) and out.dtype == torch.bfloat16 and bool(torch.isfinite(out.float()).all())
    assert abs(float(out.float().sum()) - 1.0) < 2e-2                       # softmax head
    (-torch.log(out[:, 0].float() + 1e-9)).sum().backward()
    for n, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and bool(torch.isfinite(p.grad.float()).all()), n
    # same input twice: the whole path is deterministic (no atomics anywhere)
    assert torch.equal(model(batch), out)
    del model, out
    torch.cuda.empty_cache()
    deep = _model(28)
    with torch.no_grad():
        o28 = deep(batch)
    assert bool(torch.isfinite(o28.float()).all()) and abs(float(o28.float().sum()) - 1.0) < 2e-2
    if os.environ.get("MLGNN_CANARY") != "1":          # (the guard-band allocator of a canary run keeps no statistics)
        assert torch.cuda.max_memory_allocated() < 40e9
