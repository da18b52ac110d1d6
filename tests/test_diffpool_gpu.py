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


@pytest.mark.parametrize("B,n,C,O,batched,grad_adj,normalize,bias", [
    (4, 146, 128, 37, False, False, True, True),     # level 1 pool: shared pathway adjacency
    (4, 146, 128, 32, False, False, True, True),     # level 1 embed
    (3, 37, 32, 32, True, True, True, True),         # after-pool / level 2: the adjacency is DiffPool's output
    (3, 37, 32, 10, True, True, True, False),
    (2, 10, 64, 64, True, True, True, True),
    (2, 160, 128, 64, False, False, False, True),    # limits of the fused kernel, no normalisation
    (1, 1, 1, 1, True, True, True, True),
    (5, 23, 7, 5, False, True, True, True),          # odd sizes, shared adjacency that needs a gradient
    (2, 48, 20, 33, True, True, True, True)])
def test_dense_sage_fused(B, n, C, O, batched, grad_adj, normalize, bias):
    from mlgnn.dense import dense_sage
    from mlgnn import _lib
    assert _lib.lib.mlgnn_dense_sage_supported(n, C, O, int(grad_adj)) == 1
    gen = torch.Generator().manual_seed(B * 1000 + n + O)
    x = torch.randn(B, n, C, generator=gen, requires_grad=True)
    # NOT symmetric; some rows sum below 1 (clamp active, no gradient through the degree), some above
    adj = torch.rand(*((B, n, n) if batched else (n, n)), generator=gen)
    adj = adj * (torch.arange(n)[:, None] % 3 != 0) * (2.0 / max(n, 1)) + adj * (torch.arange(n)[:, None] % 3 == 0)
    adj = adj.requires_grad_(grad_adj)
    w_rel = (torch.randn(O, C, generator=gen) * 0.3).requires_grad_(True)
    w_root = (torch.randn(O, C, generator=gen) * 0.3).requires_grad_(True)
    b = torch.randn(O, generator=gen).requires_grad_(True) if bias else None
    cot = torch.randn(B, n, O, generator=gen)
    leaves = [x, w_rel, w_root] + ([adj] if grad_adj else []) + ([b] if bias else [])
    ref = P.dense_sage_conv(x, adj, w_rel, w_root, b, normalize)
    gr = torch.autograd.grad((ref * cot).sum(), leaves)

    dev = "cuda:0"
    dl = [t.detach().to(dev).requires_grad_(True) for t in leaves]
    xd, wr, wo = dl[0], dl[1], dl[2]
    ad = dl[3] if grad_adj else adj.detach().to(dev)
    bd = dl[-1] if bias else None
    out = dense_sage(xd, ad, wr, wo, bd, normalize)
    assert type(out.grad_fn).__name__.startswith("_DenseSageFused"), "the fused path must be the one that runs"
    assert_close(out, ref, 1e-4, "dense sage fwd")
    got = torch.autograd.grad((out * cot.to(dev)).sum(), dl)
    names = ["x", "w_rel", "w_root"] + (["adj"] if grad_adj else []) + (["bias"] if bias else [])
    for name, g, r in zip(names, got, gr):
        assert_close(g, r, 1e-4, "dense sage grad " + name)


def test_dense_sage_unsupported_shapes_take_the_library_path():
    from mlgnn.dense import dense_sage
    dev = "cuda:0"
    x = torch.randn(2, 200, 16, device=dev, requires_grad=True)          # n > 160
    adj = torch.rand(200, 200, device=dev)
    w1, w2, b = torch.randn(8, 16, device=dev), torch.randn(8, 16, device=dev), torch.randn(8, device=dev)
    out = dense_sage(x, adj, w1, w2, b)
    assert_close(out, P.dense_sage_conv(x.detach().cpu(), adj.cpu(), w1.cpu(), w2.cpu(), b.cpu()), 1e-4)
    # an adjacency that needs a gradient on more than 48 nodes: not fused either
    adj2 = torch.rand(2, 60, 60, device=dev, requires_grad=True)
    x2 = torch.randn(2, 60, 16, device=dev)
    out2 = dense_sage(x2, adj2, w1, w2, b)
    out2.sum().backward()
    ref_adj = adj2.detach().cpu().requires_grad_(True)
    P.dense_sage_conv(x2.cpu(), ref_adj, w1.cpu(), w2.cpu(), b.cpu()).sum().backward()
    assert_close(adj2.grad, ref_adj.grad, 1e-4)


@pytest.mark.parametrize("B,n,C,O,batched,grad_adj", [(4, 146, 128, 32, False, False), (3, 37, 32, 10, True, True),
                                                      (2, 160, 64, 64, False, False), (5, 10, 64, 64, True, True)])
def test_dense_sage_bf16_storage_is_the_fp32_kernel_rounded_once(B, n, C, O, batched, grad_adj):
    """bf16 storage of x / adj / weights (a bf16 model's pooled levels): the same kernel arithmetic in fp32 on the same
    bf16-representable values, one rounding at each store -- the output equals the fp32 kernel's, rounded; gradients agree
    to bf16 accuracy."""
    from mlgnn.dense import dense_sage
    gen = torch.Generator().manual_seed(B + n + C)
    dev = "cuda:0"
    x = torch.randn(B, n, C, generator=gen).bfloat16()
    adj = torch.rand(*((B, n, n) if batched else (n, n)), generator=gen).bfloat16()
    wr, wo = (torch.randn(O, C, generator=gen) * 0.2).bfloat16(), (torch.randn(O, C, generator=gen) * 0.2).bfloat16()
    bias = (torch.randn(O, generator=gen) * 0.1).bfloat16()
    cot = torch.randn(B, n, O, generator=gen).bfloat16()
    res = []
    for dt in (torch.float32, torch.bfloat16):
        xd, ad = x.to(dev).to(dt).requires_grad_(True), adj.to(dev).to(dt).requires_grad_(grad_adj)
        wrd, wod, bd = (t.to(dev).to(dt).requires_grad_(True) for t in (wr, wo, bias))
        y = dense_sage(xd, ad, wrd, wod, bd, normalize=True)
        assert y.dtype == dt
        ins = [xd, wrd, wod, bd] + ([ad] if grad_adj else [])
        res.append((y,) + torch.autograd.grad(y, ins, cot.to(dev).to(dt)))
    for k, (r32, r16) in enumerate(zip(*res)):
        assert r16.dtype == torch.bfloat16
        if k == 0:                                             # y: rounded once from the fp32 result
            assert torch.equal(r16, r32.bfloat16()), k
        else:                                                  # gradients: the backward reads the SAVED y, which is the
            # rounded one here (2^-9 relative per element) -- bf16-level agreement, not bitwise
            assert float((r16.float() - r32).abs().max()) <= 2.0 ** -6 * float(r32.abs().max()), k


@pytest.mark.parametrize("B,N,K,C,batched", [(4, 146, 37, 64, False), (3, 37, 10, 64, True)])
def test_dense_diff_pool_bf16_storage_is_the_fp32_kernel_rounded_once(B, N, K, C, batched):
    """bf16 storage of z / adj / logits in the fused small-graph kernel: outputs equal the fp32 kernel's on the same
    bf16-representable values, rounded once; gradients (whose backward reads the SAVED, rounded softmax) agree to bf16
    accuracy."""
    from mlgnn.dense import dense_diff_pool
    gen = torch.Generator().manual_seed(B + N)
    dev = "cuda:0"
    z = torch.randn(B, N, C, generator=gen).bfloat16()
    s = (torch.randn(B, N, K, generator=gen) * 2).bfloat16()
    adj = torch.rand(*((B, N, N) if batched else (N, N)), generator=gen).bfloat16()
    c1, c2 = torch.randn(B, K, C, generator=gen).bfloat16(), torch.randn(B, K, K, generator=gen).bfloat16()
    res = []
    for dt in (torch.float32, torch.bfloat16):
        zd, ad, sd = (t.to(dev).to(dt).requires_grad_(True) for t in (z, adj, s))
        x, a, link, ent = dense_diff_pool(zd, ad, sd)
        assert x.dtype == dt and a.dtype == dt
        loss = (x.float() * c1.to(dev).float()).sum() + (a.float() * c2.to(dev).float()).sum() + 0.7 * link.float() + 0.3 * ent.float()
        res.append((x, a, link, ent) + torch.autograd.grad(loss, [zd, ad, sd]))
    r32, r16 = res
    assert torch.equal(r16[0], r32[0].bfloat16()) and torch.equal(r16[1], r32[1].bfloat16())
    assert abs(float(r16[2]) - float(r32[2])) <= 2.0 ** -8 * abs(float(r32[2]))
    assert abs(float(r16[3]) - float(r32[3])) <= 2.0 ** -8 * abs(float(r32[3]))
    for k in (4, 5, 6):
        assert r16[k].dtype == torch.bfloat16
        assert float((r16[k].float() - r32[k]).abs().max()) <= 2.0 ** -6 * float(r32[k].abs().max()), k
