"""Scaled split-precision (fp16 hi/lo, 3 MFMAs) tall GEMM vs fp64: layout exactness on integer data,
error statistics on random data (must be of the size of fp32 rounding, far inside the 1e-4 budget)."""
import pytest
import torch

from _util import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,R,J", [(33, 16, 32), (1000, 128, 256), (4097, 256, 128), (70000, 128, 128),
                                   (5000, 64, 64), (9999, 32, 32), (2048, 256, 64), (777, 16, 256)])
def test_layout_is_exact_on_small_integers(N, R, J):
    """Small integers are exact in fp16, so any wrong lane/row/column mapping shows up as an exact mismatch."""
    from mlgnn.dense import tall_matmul_nt, tall_matmul_supported
    assert tall_matmul_supported(N, R, J)
    gen = torch.Generator().manual_seed(N)
    a = torch.randint(-4, 5, (N, R), generator=gen).float()
    bt = torch.randint(-3, 4, (J, R), generator=gen).float()
    bt[:, 0] += torch.arange(J) % 5                              # asymmetric: a transposed tile cannot pass
    bias = torch.randint(-2, 3, (J,), generator=gen).float()
    ref = a @ bt.t() + bias
    out = tall_matmul_nt(a.cuda(), bt.cuda(), bias.cuda())
    assert torch.equal(out.cpu(), ref)
    res = torch.randint(-5, 6, (N, J), generator=gen).float()    # residual branch folded into the epilogue
    if J <= 128:
        out = tall_matmul_nt(a.cuda(), bt.cuda(), bias.cuda(), res.cuda())
        assert torch.equal(out.cpu(), ref + res)
    else:                                                        # the residual tile must fit in registers
        with pytest.raises(RuntimeError):
            tall_matmul_nt(a.cuda(), bt.cuda(), bias.cuda(), res.cuda())


@pytest.mark.parametrize("scale", [1.0, 1e-6, 3e4])
def test_split_precision_error(scale):
    from mlgnn.dense import tall_matmul_nt
    gen = torch.Generator().manual_seed(1)
    N, R, J = 20000, 256, 128
    a = torch.randn(N, R, generator=gen) * scale
    a[::7] *= 1e-3                                              # rows of very different magnitude: per-row scaling
    a[5, :] = 0.0
    a[:, 3] *= 50.0                                             # wide dynamic range inside every row
    bt = torch.randn(J, R, generator=gen) * 0.1
    ref = a.double() @ bt.double().t()
    out = tall_matmul_nt(a.cuda(), bt.cuda()).cpu().double()
    lib = (a.cuda() @ bt.cuda().t()).cpu().double()
    denom = float(ref.abs().max())
    err = float((out - ref).abs().max()) / denom
    err_lib = float((lib - ref).abs().max()) / denom
    rms = float((out - ref).pow(2).mean().sqrt()) / float(ref.pow(2).mean().sqrt())
    print("split fp16 max err %.2e (rms %.2e), fp32 library %.2e" % (err, rms, err_lib))
    # analysis in csrc/tallgemm.hip: <= 3 * 2^-22 = 7e-7 per product -- the size of fp32 rounding
    assert err < 2e-6 and rms < 1e-6


@pytest.mark.parametrize("N,R,J", [(1000, 128, 256), (4097, 256, 128), (777, 16, 32), (3000, 64, 64)])
def test_transposed_operand_is_read_in_place(N, R, J):
    """bt_transposed: the [R, J] matrix (a Linear's own weight in its input gradient) gives bit-identical results to
    the product with its contiguous transpose."""
    from mlgnn.dense import tall_matmul_nt
    gen = torch.Generator().manual_seed(R + J)
    a = torch.randn(N, R, generator=gen).cuda()
    w = (torch.randn(R, J, generator=gen) * 0.1).cuda()               # [R, J]
    ref = tall_matmul_nt(a, w.t().contiguous())
    out = tall_matmul_nt(a, w, bt_transposed=True)
    assert torch.equal(out, ref)


def test_unsupported_shapes_are_reported():
    from mlgnn.dense import tall_matmul_supported
    assert not tall_matmul_supported(1000, 128, 96)          # J not a power-of-two multiple of 32
    assert not tall_matmul_supported(1000, 100, 64)          # R not 16 * 2^k
    assert not tall_matmul_supported(1000, 512, 64)          # R beyond the supported widths
    assert not tall_matmul_supported(1000, 256, 256)         # weight image > 128 KiB


@pytest.mark.parametrize("N,R,J", [(33, 64, 64), (4097, 128, 256), (20000, 256, 128), (70001, 128, 256), (9000, 64, 128),
                                   (8191, 128, 64)])
def test_layernorm_backward_epilogue_vs_fp64(N, R, J):
    """``dA = go W`` taken through ReLU + LayerNorm backward inside the product (LN = 3 epilogue) against the same
    chain in fp64 autograd: grad_h elementwise to 1e-5 of the row's scale, d gamma / d beta (sums over N rows) to 1e-5,
    the per-row maxima exact for the grad_h the kernel wrote."""
    from mlgnn.dense import tall_matmul_ln_backward, tall_matmul_ln_backward_supported
    assert tall_matmul_ln_backward_supported(N, R, J)
    gen = torch.Generator().manual_seed(N + J)
    y1 = torch.randn(N, J, generator=gen, dtype=torch.float64) * 2 + 0.3
    gamma = (torch.rand(J, generator=gen, dtype=torch.float64) + 0.5) * torch.where(torch.rand(J, generator=gen) < 0.2, -1.0, 1.0)
    beta = torch.randn(J, generator=gen, dtype=torch.float64) * 0.3
    w = torch.randn(R, J, generator=gen, dtype=torch.float64) / R ** 0.5      # the second Linear's weight [out=R, in=J]
    go = torch.randn(N, R, generator=gen, dtype=torch.float64)
    go[::5] *= 1e-3
    eps = 1e-5
    mu, var = y1.mean(1, keepdim=True), y1.var(1, unbiased=False, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    xhat = ((y1 - mu) * rstd).float()                           # what the forward stored (fp32)
    rstd32 = rstd.squeeze(1).float()
    # fp64 reference on exactly those stored values
    xh = xhat.double().requires_grad_(True)
    g64, b64 = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a = torch.relu(xh * g64 + b64)
    (a @ w.t() * go).sum().backward()
    gxh = xh.grad                                               # gradient at xhat; LayerNorm backward from there:
    r = rstd32.double()[:, None]
    ref_h = r * (gxh - gxh.mean(1, keepdim=True) - xh.detach() * (gxh * xh.detach()).mean(1, keepdim=True))
    gh, gg, gb, rmax = tall_matmul_ln_backward(go.float().cuda(), w.float().cuda(), xhat.cuda(), rstd32.cuda(),
                                               gamma.float().cuda(), beta.float().cuda())
    gh, gg, gb, rmax = gh.cpu(), gg.cpu(), gb.cpu(), rmax.cpu()
    # entries whose pre-activation sits within rounding of the ReLU kink may fall on either side
    kink = ((xhat.double() * gamma + beta).abs() < 1e-6)
    assert int(kink.sum()) < 50
    clean = ~kink.any(1)
    scale = ref_h.abs().amax(1, keepdim=True).clamp(min=1e-30)
    err = ((gh.double() - ref_h).abs() / scale)[clean].max().item()
    assert err < 1e-5, err
    assert_close(gg[~kink.any(0)], g64.grad[~kink.any(0)].float(), 1e-5, "grad gamma")
    assert_close(gb[~kink.any(0)], b64.grad[~kink.any(0)].float(), 1e-5, "grad beta")
    assert torch.equal(rmax, gh.abs().amax(1))


def test_layernorm_backward_epilogue_equals_two_pass():
    """The fused epilogue against the two launches it replaces (tall GEMM, then LayerNorm backward) on the same inputs."""
    from mlgnn.dense import tall_matmul_ln_backward, tall_matmul_nt
    from mlgnn.norm import ln_backward_normalised
    gen = torch.Generator().manual_seed(5)
    N, R, J = 30000, 128, 256
    xhat = torch.nn.functional.layer_norm(torch.randn(N, J, generator=gen), (J,)).cuda()
    rstd = (torch.rand(N, generator=gen) + 0.5).cuda()
    gamma, beta = (torch.rand(J, generator=gen) + 0.5).cuda(), (torch.randn(J, generator=gen) * 0.2).cuda()
    w = (torch.randn(R, J, generator=gen) / R ** 0.5).cuda()
    go = torch.randn(N, R, generator=gen).cuda()
    gy = tall_matmul_nt(go, w, bt_transposed=True)
    ref = ln_backward_normalised(gy, xhat, gamma, beta, rstd, relu=True)
    out = tall_matmul_ln_backward(go, w, xhat, rstd, gamma, beta)
    for a, b, what in zip(out, ref, ("grad_h", "grad_gamma", "grad_beta", "row_max")):
        assert_close(a, b, 1e-6, what)


def test_layernorm_backward_epilogue_full_size_properties():
    """BASELINE configs[1] size (640 000 rows, 128 -> 256): the fused epilogue equals the two launches it replaces, and
    it is linear in the cotangent (the ReLU mask and the statistics depend on the stored activation only)."""
    from mlgnn.dense import tall_matmul_ln_backward, tall_matmul_nt
    from mlgnn.norm import ln_backward_normalised
    gen = torch.Generator(device="cuda").manual_seed(6)
    N, R, J = 640000, 128, 256
    xhat = torch.nn.functional.layer_norm(torch.randn(N, J, device="cuda", generator=gen), (J,))
    rstd = torch.rand(N, device="cuda", generator=gen) + 0.5
    gamma, beta = torch.rand(J, device="cuda", generator=gen) + 0.5, torch.randn(J, device="cuda", generator=gen) * 0.2
    w = torch.randn(R, J, device="cuda", generator=gen) / R ** 0.5
    g1, g2 = torch.randn(N, R, device="cuda", generator=gen), torch.randn(N, R, device="cuda", generator=gen)
    out1 = tall_matmul_ln_backward(g1, w, xhat, rstd, gamma, beta)
    ref1 = ln_backward_normalised(tall_matmul_nt(g1, w, bt_transposed=True), xhat, gamma, beta, rstd, relu=True)
    for a, b, what in zip(out1, ref1, ("grad_h", "grad_gamma", "grad_beta", "row_max")):
        assert_close(a, b, 2e-6, what)
    out2 = tall_matmul_ln_backward(g2, w, xhat, rstd, gamma, beta)
    mix = tall_matmul_ln_backward(0.5 * g1 - 2.0 * g2, w, xhat, rstd, gamma, beta)
    assert_close(mix[0], 0.5 * out1[0] - 2.0 * out2[0], 2e-6, "linearity of grad_h")
    assert_close(mix[1], 0.5 * out1[1] - 2.0 * out2[1], 1e-5, "linearity of grad_gamma")
    assert_close(mix[2], 0.5 * out1[2] - 2.0 * out2[2], 1e-5, "linearity of grad_beta")
