"""bf16 activation storage (BASELINE configs[4] dtype) of the aggregation kernels: inputs rounded to
bf16, fp32 arithmetic inside, outputs rounded to bf16.  Checked against the fp32 oracle on the
bf16-rounded inputs with a bf16-sized tolerance (3 significant digits: 2^-8 relative)."""
import pytest
import torch

from _util import assert_close
from oracle import gcn_lib as G

pytestmark = pytest.mark.gpu
TOL = 2e-2


@pytest.mark.parametrize("aggr", ["softmax", "max", "mean", "add", "power"])
@pytest.mark.parametrize("edge_kind", ["rank1", "full", "none"])
@pytest.mark.parametrize("d", [64, 100, 256])
def test_bf16_aggregate(aggr, edge_kind, d):
    from mlgnn import CSRGraph, RankOneEdge, gen_aggregate
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(d)
    N, E = 400, 5000
    ei = torch.stack([torch.randint(0, N, (E,), generator=gen), torch.randint(0, N - 2, (E,), generator=gen)])
    rb = lambda t: t.to(torch.bfloat16).float()                       # what the kernel will see
    x = rb(torch.randn(N, d, generator=gen))
    a = torch.rand(E, generator=gen)
    u, v = torch.randn(d, generator=gen) * 0.5, torch.randn(d, generator=gen) * 0.2
    ef = rb(torch.randn(E, d, generator=gen) * 0.5)
    cot = rb(torch.randn(N, d, generator=gen))
    leaves = {"x": x.clone().requires_grad_(True)}
    if edge_kind == "rank1":
        leaves["u"], leaves["v"] = u.clone().requires_grad_(True), v.clone().requires_grad_(True)
        e = a[:, None] * leaves["u"] + leaves["v"]
    elif edge_kind == "full":
        leaves["ef"] = ef.clone().requires_grad_(True)
        e = leaves["ef"]
    else:
        e = 0
    msg = torch.relu(leaves["x"][ei[0]] + e) + 1e-7
    ref = G.gen_aggregate(msg, ei[1], N, aggr, t=1.0, p=2.0)
    names = list(leaves)
    ref_g = dict(zip(names, torch.autograd.grad((ref * cot).sum(), [leaves[k] for k in names])))

    gl = {k: (val.detach().to(dev).to(torch.bfloat16 if k in ("x", "ef") else torch.float32)).requires_grad_(True)
          for k, val in leaves.items()}
    graph = CSRGraph(ei.to(dev), N)
    edge = RankOneEdge(a.to(dev), gl["u"], gl["v"]) if edge_kind == "rank1" else (gl["ef"] if edge_kind == "full" else None)
    out = gen_aggregate(gl["x"], graph, edge, aggr=aggr, t=1.0, p=2.0)
    assert out.dtype == torch.bfloat16
    assert_close(out.float(), ref, TOL, "bf16 %s/%s fwd" % (aggr, edge_kind))
    got = torch.autograd.grad((out.float() * cot.to(dev)).sum(), [gl[k] for k in names])
    for k, g in zip(names, got):
        assert g.dtype == gl[k].dtype
        assert_close(g.float(), ref_g[k], TOL, "bf16 %s/%s grad %s" % (aggr, edge_kind, k))


def test_bf16_matches_fp32_kernel_on_rounded_inputs():
    """Same arithmetic, different storage: the bf16 path equals the fp32 path up to the final rounding."""
    from mlgnn import CSRGraph, RankOneEdge, gen_aggregate
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    N, E, d = 3000, 48000, 256
    ei = torch.randint(0, N, (2, E), generator=gen).to(dev)
    g = CSRGraph(ei, N)
    x = torch.randn(N, d, generator=gen).to(dev).to(torch.bfloat16)
    a = torch.rand(E, generator=gen).to(dev)
    u, v = torch.randn(d, generator=gen).to(dev) * 0.3, torch.randn(d, generator=gen).to(dev) * 0.1
    lo = gen_aggregate(x, g, RankOneEdge(a, u, v), aggr="softmax", add_root=True)
    hi = gen_aggregate(x.float(), g, RankOneEdge(a, u, v), aggr="softmax", add_root=True)
    # the lane-group split differs (8 vs 4 channels per lane), so fp32 sums may differ in the last bit
    # before the rounding: allow one bf16 ulp (2^-8 relative)
    assert_close(lo.float(), hi, 2 ** -7, "bf16 storage vs fp32 storage")
    assert float((lo != hi.to(torch.bfloat16)).float().mean()) < 0.01


# ---- dense path in bf16 storage: tall GEMM (csrc/tallgemm_bf16.hip) and LayerNorm (csrc/norm.hip) --------------

@pytest.mark.parametrize("N,R,J", [(33, 16, 32), (1000, 128, 256), (4097, 256, 512), (9000, 512, 256), (5000, 64, 64),
                                   (777, 32, 96), (2048, 48, 160), (70000, 256, 128), (100, 1024, 64)])
def test_bf16_tall_gemm_layout_is_exact_on_small_integers(N, R, J):
    """Small integers and their dot products (< 2^8 in magnitude after the bias) are exact in bf16, so any wrong
    lane / row / column / slice mapping -- including the pair-interleaved output columns -- is an exact mismatch."""
    from mlgnn.dense import tall_matmul_nt, tall_matmul_supported
    assert tall_matmul_supported(N, R, J, torch.bfloat16)
    gen = torch.Generator().manual_seed(N + J)
    a = torch.zeros(N, R)
    hot = torch.randint(0, R, (N, 3), generator=gen)                  # three non-zeros per row: sums stay small
    a.scatter_(1, hot, torch.randint(-3, 4, (N, 3), generator=gen).float())
    bt = torch.randint(-4, 5, (J, R), generator=gen).float()
    bt[:, 0] += torch.arange(J) % 5                                   # asymmetric: a transposed tile cannot pass
    bias = torch.randint(-2, 3, (J,), generator=gen).float()
    ref = a @ bt.t() + bias
    assert float(ref.abs().max()) < 256
    bf = lambda t: t.cuda().to(torch.bfloat16)
    out = tall_matmul_nt(bf(a), bf(bt), bias.cuda())
    assert out.dtype == torch.bfloat16 and torch.equal(out.float().cpu(), ref)
    res = torch.randint(-5, 6, (N, J), generator=gen).float()
    out = tall_matmul_nt(bf(a), bf(bt), bias.cuda(), bf(res))
    assert float((ref + res).abs().max()) < 256 and torch.equal(out.float().cpu(), ref + res)
    out = tall_matmul_nt(bf(a), bf(bt))                               # no bias
    assert torch.equal(out.float().cpu(), a @ bt.t())


def test_bf16_tall_gemm_accumulates_in_fp32():
    """Random data: the only rounding is the final store (relative 2^-9 of each result), not one per product."""
    from mlgnn.dense import tall_matmul_nt
    gen = torch.Generator().manual_seed(5)
    N, R, J = 20000, 512, 256
    a = torch.randn(N, R, generator=gen).to(torch.bfloat16)
    bt = (torch.randn(J, R, generator=gen) * 0.1).to(torch.bfloat16)
    ref = a.double() @ bt.double().t()
    out = tall_matmul_nt(a.cuda(), bt.cuda()).cpu().double()
    rel = ((out - ref).abs() / ref.abs().clamp(min=1e-2)).max()
    assert float(rel) < 2.0 ** -8, float(rel)                         # one bf16 rounding (2^-9) + fp32 accumulation noise
    assert not tall_matmul_supported_odd()


def tall_matmul_supported_odd():
    from mlgnn.dense import tall_matmul_supported
    return (tall_matmul_supported(1000, 40, 64, torch.bfloat16) or tall_matmul_supported(1000, 64, 48, torch.bfloat16)
            or tall_matmul_supported(1000, 2048, 64, torch.bfloat16))


@pytest.mark.parametrize("rows,d", [(1, 8), (7, 16), (1000, 64), (2049, 104), (4097, 256), (3000, 512), (70000, 128)])
@pytest.mark.parametrize("relu", [False, True])
def test_bf16_layer_norm_act(rows, d, relu):
    """bf16 storage, fp32 statistics: against fp32 LayerNorm on the bf16-rounded input.  The forward differs by the
    output rounding only; gradients by the roundings of grad_x (parameter gradients are fp32 sums: tight)."""
    import torch.nn.functional as F
    from mlgnn.norm import fused_supported, layer_norm_act, layer_norm_act_fork
    gen = torch.Generator().manual_seed(rows + d)
    rb = lambda t: t.to(torch.bfloat16).float()
    x = rb(torch.randn(rows, d, generator=gen) * 2 + 0.5).requires_grad_(True)
    w = rb(torch.rand(d, generator=gen) + 0.5).requires_grad_(True)
    b = rb(torch.randn(d, generator=gen) * 0.3).requires_grad_(True)
    cot, cot2 = rb(torch.randn(rows, d, generator=gen)), rb(torch.randn(rows, d, generator=gen))
    pre = F.layer_norm(x, (d,), w, b, 1e-5)
    ref = F.relu(pre) if relu else pre
    gr = torch.autograd.grad((ref * cot).sum() + (x * cot2).sum(), [x, w, b])
    # the ReLU mask is recomputed from x in the backward: where the pre-activation is within rounding of zero the
    # two implementations may legitimately disagree on its sign (a handful of the 9 M elements of the largest case)
    keep = (pre.detach().abs() > 1e-4) if relu else torch.ones_like(pre, dtype=torch.bool)
    dev = "cuda:0"
    xd, wd, bd = (t.detach().to(dev).to(torch.bfloat16).requires_grad_(True) for t in (x, w, b))
    assert fused_supported(xd)
    out, ident = layer_norm_act_fork(xd, wd, bd, 1e-5, relu)
    assert out.dtype == torch.bfloat16
    assert_close(out.float(), ref, 1e-2, "bf16 ln fwd")
    got = torch.autograd.grad((out.float() * cot.to(dev)).sum() + (ident.float() * cot2.to(dev)).sum(), [xd, wd, bd])
    assert all(g.dtype == torch.bfloat16 for g in got)
    assert_close(got[0].float().cpu() * keep, gr[0] * keep, 1e-2, "bf16 ln grad x (+ identity branch)")
    assert_close(got[1].float(), gr[1], 1e-2, "bf16 ln grad gamma")
    assert_close(got[2].float(), gr[2], 1e-2, "bf16 ln grad beta")
    out2 = layer_norm_act(xd, wd, bd, 1e-5, relu)
    assert torch.equal(out2, out)


@pytest.mark.parametrize("N,K,M,res", [(9000, 256, 512, False), (8200, 512, 256, True), (10000, 128, 128, True)])
def test_bf16_linear_forward_backward(N, K, M, res):
    """mlgnn.dense.linear on bf16 tensors: native forward / input gradient, library weight gradient; against fp32."""
    from mlgnn.dense import linear
    gen = torch.Generator().manual_seed(N)
    rb = lambda t: t.to(torch.bfloat16).float()
    x = rb(torch.randn(N, K, generator=gen)).requires_grad_(True)
    w = rb(torch.randn(M, K, generator=gen) * K ** -0.5).requires_grad_(True)
    b = rb(torch.randn(M, generator=gen) * 0.1).requires_grad_(True)
    r = rb(torch.randn(N, M, generator=gen)).requires_grad_(True)
    cot = rb(torch.randn(N, M, generator=gen) * 0.1)
    ref = x @ w.t() + b + (r if res else 0)
    gr = torch.autograd.grad((ref * cot).sum(), [x, w, b] + ([r] if res else []))
    dev = "cuda:0"
    xd, wd, bd, rd = (t.detach().to(dev).to(torch.bfloat16).requires_grad_(True) for t in (x, w, b, r))
    out = linear(xd, wd, bd, rd if res else None)
    assert out.dtype == torch.bfloat16 and out.grad_fn.name().startswith("_TallLinear")
    assert_close(out.float(), ref, 1e-2, "bf16 linear fwd")
    got = torch.autograd.grad((out.float() * cot.to(dev)).sum(), [xd, wd, bd] + ([rd] if res else []))
    for name, g, e in zip(("x", "w", "b", "res"), got, gr):
        assert g.dtype == torch.bfloat16
        assert_close(g.float(), e, 2e-2, "bf16 linear grad " + name)


def test_bf16_deepergcn_runs_on_native_kernels_and_tracks_fp32():
    """configs[4]-shaped (small) DeeperGCN in bf16: every tall Linear / LayerNorm of the layer stack goes through
    the HIP kernels (no ATen layer_norm / addmm on [N, .] tensors), and the prediction tracks the fp32 model."""
    from types import SimpleNamespace
    from _util import make_args
    from models import get_model
    from torch.profiler import ProfilerActivity, profile
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(11)
    N, E, H = 12000, 90000, 256
    ei = torch.randint(0, N, (2, E), generator=gen)
    mk = lambda dt: SimpleNamespace(x=torch.randn(N, 3, generator=torch.Generator().manual_seed(1)).to(dev).to(dt),
                                    edge_index=ei.to(dev),
                                    edge_attr=torch.rand(E, 1, generator=torch.Generator().manual_seed(2)).to(dev),
                                    batch=(torch.arange(N) // (N // 4)).clamp(max=3).to(dev), age=torch.zeros(4, device=dev, dtype=dt),
                                    pathway_node_attr=None, node_size=torch.full((4,), N // 4, device=dev))
    args = make_args(num_layers=3, hidden_channels=H, dropout=0.0, conv_encode_edge=True, use_edge_attr=True,
                     use_column="w", global_edge="none", gcn_aggr="softmax", block="res+", norm="layer",
                     graph_pooling="mean", pathway_readout=None)
    torch.manual_seed(0)
    model = get_model("deepergcn")(args).to(dev)
    ref = model(mk(torch.float32)).detach()
    model.to(torch.bfloat16)
    batch = mk(torch.bfloat16)
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        out = model(batch)
        (-torch.log(out[:, 0].float() + 1e-9)).sum().backward()
    names = {e.key for e in prof.key_averages()}
    assert "aten::native_layer_norm" not in names and "aten::native_layer_norm_backward" not in names, names
    assert out.dtype == torch.bfloat16
    assert_close(out.float(), ref, 5e-2, "bf16 model vs fp32 model")
    assert all(p.grad is not None and bool(torch.isfinite(p.grad.float()).all()) for p in model.parameters() if p.requires_grad)


@pytest.mark.parametrize("N,M,K", [(64, 64, 128), (1000, 512, 256), (4129, 256, 512), (33, 128, 128), (9001, 192, 384),
                                   (70000, 256, 256), (1, 64, 128)])
def test_bf16_wgrad_is_exact_on_small_integers(N, M, K):
    """dW = go^T x, db = colsum(go) through the transposed LDS reads: sparse small-integer operands keep every sum
    exactly representable in fp32, so a wrong row / column / k-half mapping or a lost ragged tail is a mismatch."""
    from mlgnn.dense import _wgrad
    gen = torch.Generator().manual_seed(N + M)
    go = torch.zeros(N, M)
    go.scatter_(1, torch.randint(0, M, (N, 5), generator=gen), torch.randint(-3, 4, (N, 5), generator=gen).float())
    x = torch.randint(-4, 5, (N, K), generator=gen).float()
    x[:, 0] += torch.arange(N) % 3
    ref_w, ref_b = go.t() @ x, go.sum(0)
    gw, gb = _wgrad(go.cuda().to(torch.bfloat16), x.cuda().to(torch.bfloat16))
    assert gw.dtype == torch.float32 and gw.shape == (M, K)
    assert torch.equal(gw.cpu(), ref_w) and torch.equal(gb.cpu(), ref_b)


def test_bf16_wgrad_of_an_empty_batch_is_zero():
    from mlgnn.dense import _wgrad
    gw, gb = _wgrad(torch.empty(0, 128, device="cuda:0", dtype=torch.bfloat16),
                    torch.empty(0, 256, device="cuda:0", dtype=torch.bfloat16))
    assert gw.shape == (128, 256) and not bool(gw.any()) and not bool(gb.any())


def test_bf16_wgrad_random_matches_fp64_and_is_deterministic():
    from mlgnn.dense import _wgrad
    gen = torch.Generator().manual_seed(9)
    N, M, K = 50000, 512, 256
    go = (torch.randn(N, M, generator=gen) * 0.1).to(torch.bfloat16)
    x = torch.randn(N, K, generator=gen).to(torch.bfloat16)
    ref = go.double().t() @ x.double()
    gw, gb = _wgrad(go.cuda(), x.cuda())
    gw2, gb2 = _wgrad(go.cuda(), x.cuda())
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)                       # fixed summation order
    err = float((gw.cpu().double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 1e-5, err                                                     # exact products, fp32 accumulation
    assert_close(gb.cpu(), go.double().sum(0), 1e-5, "bias gradient")


@pytest.mark.parametrize("rows,d", [(1000, 128), (4097, 256), (33, 512), (70000, 64)])
def test_bf16_msg_norm_add(rows, d):
    """MsgNorm + root add (torch_message.py:175-179, torch_vertex.py:86-89) with bf16 storage on the HIP kernel: fp32
    arithmetic inside, one rounding per output; reference = fp32 on the bf16-rounded inputs."""
    import torch.nn.functional as F
    from mlgnn.norm import msg_norm_add
    from torch.profiler import ProfilerActivity, profile
    gen = torch.Generator().manual_seed(rows + d)
    rb = lambda t: t.to(torch.bfloat16).float()
    x = rb(torch.randn(rows, d, generator=gen)).requires_grad_(True)
    m = rb(torch.rand(rows, d, generator=gen) * 3).requires_grad_(True)
    with torch.no_grad():
        m[0] = 0.0
    s = torch.tensor([0.7], requires_grad=True)
    cot = rb(torch.randn(rows, d, generator=gen))
    ref = x + F.normalize(m, p=2.0, dim=1) * x.norm(p=2, dim=1, keepdim=True) * s
    gr = torch.autograd.grad((ref * cot).sum(), [x, m, s])
    dev = "cuda:0"
    xd, md = (t.detach().to(dev).to(torch.bfloat16).requires_grad_(True) for t in (x, m))
    sd = s.detach().to(dev).requires_grad_(True)
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        out = msg_norm_add(xd, md, sd)
        got = torch.autograd.grad((out.float() * cot.to(dev)).sum(), [xd, md, sd])
    assert "aten::linalg_vector_norm" not in {e.key for e in prof.key_averages()}      # the HIP kernel ran, not ATen
    assert out.dtype == torch.bfloat16
    assert_close(out.float(), ref, 2.0 ** -8, "bf16 msgnorm fwd", elementwise=True)
    for name, g, r in zip(("x", "m", "scale"), got, gr):
        assert_close(g.float(), r, 2.0 ** -7, "bf16 msgnorm grad " + name)


def test_bf16_pooled_levels_run_on_the_native_kernels():
    """A bf16 model's small pooled levels (DenseSAGE, DiffPool <= 160 nodes, projection pooling) go through the
    hand-written fp32 kernels behind casts instead of ATen formulas; numbers = fp32 oracle on the rounded inputs."""
    from mlgnn.dense import dense_diff_pool, dense_sage
    from oracle import primitives as OP
    gen = torch.Generator().manual_seed(3)
    rb = lambda t: t.to(torch.bfloat16).float()
    B, n, C, O, K = 6, 146, 32, 32, 37
    x = rb(torch.randn(B, n, C, generator=gen))
    adj = rb(torch.rand(n, n, generator=gen))
    wr, wo = rb(torch.randn(O, C, generator=gen) * 0.2), rb(torch.randn(O, C, generator=gen) * 0.2)
    b = rb(torch.randn(O, generator=gen) * 0.1)
    dev = "cuda:0"

    y = dense_sage(x.to(dev).bfloat16(), adj.to(dev).bfloat16(), wr.to(dev).bfloat16(), wo.to(dev).bfloat16(), b.to(dev).bfloat16())
    ref = OP.dense_sage_conv(x, adj, wr, wo, b, normalize=True)
    assert y.dtype == torch.bfloat16
    assert_close(y.float(), ref, 2.0 ** -7, "bf16 DenseSAGE")
    s = rb(torch.randn(B, n, K, generator=gen))
    z = rb(torch.randn(B, n, C, generator=gen))
    ox, oa, ol, oe = dense_diff_pool(z.to(dev).bfloat16(), adj.to(dev).bfloat16(), s.to(dev).bfloat16())
    rx, ra, rl, re = OP.dense_diff_pool(z, adj, s)
    assert ox.dtype == torch.bfloat16 and oa.dtype == torch.bfloat16
    assert_close(ox.float(), rx, 2.0 ** -7, "bf16 small DiffPool x")
    assert_close(oa.float(), ra, 2.0 ** -7, "bf16 small DiffPool adj")
    assert abs(float(ol) - float(rl)) <= 2.0 ** -7 * float(rl) and abs(float(oe) - float(re)) <= 2.0 ** -7 * abs(float(re))


@pytest.mark.parametrize("N,M,K", [(4099, 512, 256), (33, 128, 64), (9000, 256, 128)])
def test_bf16_input_gradient_with_shifted_cotangent(N, M, K):
    """csrc/tallgemm_bf16.hip SHIFT: the input gradient of a Linear behind a softmax aggregation and that aggregation's
    rescaled cotangent from one epilogue -- c bitwise equal to the plain product, gt = bf16(c * 2^(-lse)) from the
    rounded c (what the streaming pre-pass of csrc/aggregate_bwd.hip computes), the flag raised past |lse| = 60."""
    from mlgnn import dense as D
    g = torch.Generator(device="cuda:0").manual_seed(N)
    go = torch.randn(N, M, device="cuda:0", generator=g).bfloat16()
    w = (torch.randn(M, K, device="cuda:0", generator=g) * 0.1).bfloat16()
    lse = torch.randn(N, K, device="cuda:0", generator=g) * 6.0
    gx, gt, flag = D.tall_matmul_bf16_shift(go, w, lse)
    plain = D.tall_matmul_nt(go, w, bt_transposed=True)
    assert torch.equal(gx, plain)
    want = (gx.float() * torch.exp2(-lse)).bfloat16()
    assert torch.equal(gt, want)
    assert int(flag[0]) == 0
    lse[N // 2, 3] = -61.0
    _, _, flag = D.tall_matmul_bf16_shift(go, w, lse)
    assert int(flag[0]) == 1


def test_bf16_parameter_copies_are_never_stale():
    """The fp32 copies the kernels read of a bf16 model's [d]-sized parameters (mlgnn.ops.f32_cached): an edit through
    ``.data`` does not bump the version counter, so (a) a parameter no optimizer has stepped is cast per call, and (b) one
    an optimizer steps is re-cast after every optimizer step -- also when that optimizer writes through
    ``p.data.copy_`` as the reference's utils/optim.py (RAdam, AdamW) does -- and after a storage change."""
    from mlgnn.norm import layer_norm_act
    from mlgnn import ops
    DEV = "cuda:0"
    g = torch.Generator(device=DEV).manual_seed(4)
    x = torch.randn(9000, 256, device=DEV, generator=g).to(torch.bfloat16)
    ln = torch.nn.LayerNorm(256).to(DEV).to(torch.bfloat16)
    with torch.no_grad():
        y0 = layer_norm_act(x, ln.weight, ln.bias, ln.eps, True).float()
        ln.weight.data.copy_(torch.full_like(ln.weight, 3.0))          # (a) never stepped: version unchanged, still seen
        y1 = layer_norm_act(x, ln.weight, ln.bias, ln.eps, True).float()
    assert not torch.equal(y0, y1)

    class DataCopySGD(torch.optim.Optimizer):                          # updates like utils/optim.py: p.data.copy_(...)
        def __init__(self, params):
            super().__init__(params, dict(lr=0.5))

        def step(self):
            for grp in self.param_groups:
                for p in grp["params"]:
                    if p.grad is not None:
                        p.data.copy_((p.data.float() - grp["lr"] * p.grad.float()).to(p.dtype))

    opt = DataCopySGD(ln.parameters())
    outs = []
    for _ in range(3):
        opt.zero_grad()
        y = layer_norm_act(x, ln.weight, ln.bias, ln.eps, True)
        y.float().square().mean().backward()
        v = ln.weight._version
        opt.step()
        assert ln.weight._version == v                                 # the update did not bump the version ...
        outs.append(y.detach().float())
    assert getattr(ln.weight, "_mlgnn_stepped", False)
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])   # ... and was seen all the same
    with torch.no_grad():
        ref = torch.relu(torch.nn.functional.layer_norm(x.float(), (256,), ln.weight.float(), ln.bias.float(), ln.eps))
        got = layer_norm_act(x, ln.weight, ln.bias, ln.eps, True).float()
        assert float((got - ref).abs().max()) <= 0.05 * float(ref.abs().max())
        # (b') between optimizer steps the copy is kept: same object twice
        assert ops.f32_cached(ln.weight) is ops.f32_cached(ln.weight)
        ln.weight.data = ln.weight.data.clone()                        # a storage change is seen
        c1 = ops.f32_cached(ln.weight)
        ln.weight.data.mul_(2.0)
        ops.invalidate_param_cache()                                   # the documented way for a manual in-place edit
        assert not torch.equal(c1, ops.f32_cached(ln.weight))
