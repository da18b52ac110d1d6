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
