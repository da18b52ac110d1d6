"""Fused fp32-MFMA DiffPool contraction vs the oracle's dense_diff_pool (PyG 2.2.0 formula)."""
import pytest
import torch

from _util import assert_close
from oracle import primitives as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,N,K,C,batched", [(4, 146, 37, 32, False), (3, 37, 10, 64, True), (2, 160, 48, 64, False),
                                             (1, 5, 2, 3, False), (5, 17, 16, 16, True), (6, 146, 37, 8, False),
                                             (2, 33, 47, 1, True)])
def test_dense_diff_pool_fused(B, N, K, C, batched):
    from mlgnn.dense import dense_diff_pool
    from mlgnn import _lib
    assert _lib.lib.mlgnn_diffpool_fwd_supported(N, K, C) == 1
    gen = torch.Generator().manual_seed(B * 1000 + N)
    z = torch.randn(B, N, C, generator=gen, requires_grad=True)
    s = (torch.randn(B, N, K, generator=gen) * 2).requires_grad_(True)
    # deliberately NOT symmetric: a transposed operand read must fail
    adj = (torch.rand(*((B, N, N) if batched else (N, N)), generator=gen) + 0.1 * torch.arange(N)[None, :] / N)
    adj = adj.requires_grad_(True)
    c1 = torch.randn(B, K, C, generator=gen)
    c2 = torch.randn(B, K, K, generator=gen)
    x1, a1, l1, e1 = P.dense_diff_pool(z, adj, s)
    gr = torch.autograd.grad((x1 * c1).sum() + (a1 * c2).sum() + 0.7 * l1 + 0.3 * e1, [z, adj, s])

    dev = "cuda:0"
    zd, ad, sd = (t.detach().to(dev).requires_grad_(True) for t in (z, adj, s))
    x2, a2, l2, e2 = dense_diff_pool(zd, ad, sd)
    assert_close(x2, x1, 1e-4, "S^T Z")
    assert_close(a2, a1, 1e-4, "S^T A S")
    assert_close(l2, l1, 1e-4, "link")
    assert_close(e2, e1, 1e-4, "entropy")
    got = torch.autograd.grad((x2 * c1.to(dev)).sum() + (a2 * c2.to(dev)).sum() + 0.7 * l2 + 0.3 * e2, [zd, ad, sd])
    for name, g, r in zip(("z", "adj", "s"), got, gr):
        assert_close(g, r, 1e-4, "diffpool grad " + name)


def test_large_pooled_graphs_take_library_gemms():
    from mlgnn.dense import dense_diff_pool
    from mlgnn import _lib
    assert _lib.lib.mlgnn_diffpool_fwd_supported(512, 128, 64) == 0
    dev = "cuda:0"
    z, s = torch.randn(2, 200, 16, device=dev), torch.randn(2, 200, 50, device=dev)
    adj = torch.rand(200, 200, device=dev)
    x, a, l, e = dense_diff_pool(z, adj, s)
    xr, ar, lr, er = P.dense_diff_pool(z.cpu(), adj.cpu(), s.cpu())
    assert_close(x, xr, 1e-4)
    assert_close(a, ar, 1e-4)
    assert_close(l, lr, 1e-4)
    assert_close(e, er, 1e-4)
