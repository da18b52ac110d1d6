"""[N, d] tensors of 4 GiB and more in the aggregation kernels (64-bit row addresses: the WIDE instantiations of
csrc/aggregate_fwd.hip / aggregate_bwd.hip).  Reference semantics are those of the other aggregation tests
(models/gcn_lib/sparse/torch_vertex.py:94-101, torch_message.py:44-85); what is checked here is that the WIDE kernels
(a) give the bits of the 32-bit kernels on inputs both can take (``MLGNN_FORCE_WIDE=1`` in a child process) and (b) are
right past the 4 GiB line, against a direct evaluation of sampled rows."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
from mlgnn import CSRGraph, RankOneEdge, gen_aggregate
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
n, e = 5000, 70000
ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)]).to(dev)
graph = CSRGraph(ei, n)
out = {}
for dt in (torch.float32, torch.bfloat16):
    for d in (128, 64):
        x0 = torch.randn(n, d, generator=g).to(dev).to(dt)
        go = torch.randn(n, d, generator=g).to(dev).to(dt)
        edge = RankOneEdge(torch.rand(e, generator=g).to(dev), (torch.randn(d, 1, generator=g) * 0.3).to(dev),
                           (torch.randn(d, generator=g) * 0.1).to(dev))
        for aggr in ("softmax", "max", "mean"):
            x = x0.clone().requires_grad_(True)
            y = gen_aggregate(x, graph, edge, aggr=aggr, add_root=True)
            y.backward(go)
            out["%%s/%%d/%%s" %% (dt, d, aggr)] = (y.detach().float().cpu(), x.grad.float().cpu())
torch.save(out, sys.argv[1])
""" % os.path.join(ROOT, "multilevel-gnn_amd")


def test_wide_kernels_give_the_bits_of_the_32_bit_kernels(tmp_path):
    outs = []
    for force in ("0", "1"):
        path = str(tmp_path / ("o%s.pt" % force))
        env = dict(os.environ, MLGNN_FORCE_WIDE=force)
        env.pop("MLGNN_CANARY", None)
        subprocess.run([sys.executable, "-c", _CHILD, path], check=True, env=env, timeout=600)
        outs.append(torch.load(path, weights_only=True))
    a, b = outs
    assert a.keys() == b.keys() and len(a) == 12
    for k in a:
        assert torch.equal(a[k][0], b[k][0]) and torch.equal(a[k][1], b[k][1]), k


def test_rows_past_4_gib_against_direct_evaluation():
    """N = 8 500 000 nodes x d = 128 fp32 = 4.35 GB per tensor: mean aggregation with the rank-1 edge term and the root
    add; sampled destination rows (the last rows of the tensor among them) recomputed with torch on the device."""
    from mlgnn import CSRGraph, RankOneEdge, gen_aggregate
    dev = torch.device("cuda:0")
    free, _ = torch.cuda.mem_get_info()
    if free < 60e9:
        pytest.skip("needs ~60 GB of free HBM")
    g = torch.Generator(device=dev).manual_seed(5)
    n, e, d = 8_500_000, 12_000_000, 128
    assert n * d * 4 >= 2 ** 32
    src = torch.randint(0, n, (e,), generator=g, device=dev)
    dst = torch.randint(0, n, (e,), generator=g, device=dev)
    dst[:4000] = n - 1 - (torch.arange(4000, device=dev) % 7)          # edges into the very last rows
    src[:4000] = n - 1 - (torch.arange(4000, device=dev) % 1000)       # ... from rows past the 4 GiB line
    graph = CSRGraph(torch.stack([src, dst]), n)
    x = torch.randn(n, d, generator=g, device=dev).requires_grad_(True)
    a = torch.rand(e, generator=g, device=dev)
    u = (torch.randn(d, 1, generator=g, device=dev) * 0.3)
    v = (torch.randn(d, generator=g, device=dev) * 0.1)
    y = gen_aggregate(x, graph, RankOneEdge(a, u, v), aggr="mean", add_root=True, eps=1e-7)
    rows = torch.cat([torch.arange(n - 8, n, device=dev), torch.randint(0, n, (24,), generator=g, device=dev)])
    for r in rows.tolist():
        sel = (dst == r).nonzero().reshape(-1)
        m = torch.relu(x.detach()[src[sel]] + a[sel, None] * u[:, 0][None, :] + v[None, :]) + 1e-7
        ref = x.detach()[r] + (m.sum(0) / max(len(sel), 1) if len(sel) else 0.0)
        assert float((y[r].detach() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), r
    # backward: grad wrt x of sum(y * w) at sampled source rows, w nonzero only on a few destination rows
    w = torch.zeros(n, d, device=dev)
    hot = torch.arange(n - 7, n, device=dev)
    w[hot] = torch.randn(7, d, generator=g, device=dev)
    y.backward(w)
    for j in src[:40].unique().tolist():
        sel = ((src == j) & (dst >= n - 7)).nonzero().reshape(-1)
        z = x.detach()[j][None, :] + a[sel, None] * u[:, 0][None, :] + v[None, :]
        deg = torch.stack([(dst == int(t)).sum() for t in dst[sel]]).clamp(min=1).float()
        ref = ((z > 0).float() * w[dst[sel]] / deg[:, None]).sum(0) + w[j]
        assert float((x.grad[j] - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), j
