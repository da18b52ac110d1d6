"""Per-graph readout kernels vs the oracle's global_pool (torch_scatter semantics)."""
import pytest
import torch

from _util import assert_close
from oracle import primitives as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["sum", "mean", "max"])
@pytest.mark.parametrize("d", [3, 4, 32, 100, 101, 128, 320])
def test_global_pool(kind, d):
    from mlgnn.pool import global_pool
    gen = torch.Generator().manual_seed(d)
    sizes = [1, 0, 777, 5, 3000, 64, 0]                        # empty graphs in the middle and at the end
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    N = int(batch.numel())
    x = torch.randn(N, d, generator=gen, requires_grad=True)
    with torch.no_grad():
        x[10:20] = x[10]                                       # ties for max: the first row must win
    cot = torch.randn(len(sizes), d, generator=gen)
    ref = P.global_pool(x, batch, kind, len(sizes))
    (gr,) = torch.autograd.grad((ref * cot).sum(), [x])
    dev = "cuda:0"
    xd = x.detach().to(dev).requires_grad_(True)
    out = global_pool(xd, batch.to(dev), kind, len(sizes))
    assert_close(out, ref, 1e-4, "pool fwd " + kind)
    (g,) = torch.autograd.grad((out * cot.to(dev)).sum(), [xd])
    assert_close(g, gr, 1e-4, "pool grad " + kind)
