"""Split-row weight/bias gradient kernel (three-term bf16 split on MFMA, fp32 accumulation) vs torch
autograd on the CPU."""
import pytest
import torch
import torch.nn.functional as F

from _util import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,K,M", [(10000, 128, 256), (9001, 256, 128), (8200, 3, 128), (8193, 100, 60),
                                   (20011, 32, 32), (8192, 1, 1), (8195, 130, 33), (70000, 128, 128),
                                   (8192, 512, 64), (8192, 64, 512)])
@pytest.mark.parametrize("bias", [True, False])
def test_tall_linear_gradients(N, K, M, bias):
    from mlgnn.dense import linear
    gen = torch.Generator().manual_seed(N + K + M)
    x = torch.randn(N, K, generator=gen, requires_grad=True)
    w = (torch.randn(M, K, generator=gen) * 0.1).requires_grad_(True)
    b = torch.randn(M, generator=gen).requires_grad_(True) if bias else None
    cot = torch.randn(N, M, generator=gen)
    # asymmetric integer-valued probe in a few rows: a transposed or permuted tile cannot pass
    with torch.no_grad():
        cot[:64] = torch.arange(64 * M, dtype=torch.float32).reshape(64, M) % 7 - 3
    ref = F.linear(x, w, b)
    leaves = [x, w] + ([b] if bias else [])
    gr = torch.autograd.grad((ref * cot).sum(), leaves)
    dev = "cuda:0"
    dl = [t.detach().to(dev).requires_grad_(True) for t in leaves]
    out = linear(dl[0], dl[1], dl[2] if bias else None)
    assert_close(out, ref, 1e-4, "linear fwd")
    got = torch.autograd.grad((out * cot.to(dev)).sum(), dl)
    for name, g, r in zip(("x", "weight", "bias"), got, gr):
        assert_close(g, r, 1e-4, "linear grad " + name)


@pytest.mark.parametrize("scale", [1.0, 1e-12, 1e9])
def test_wgrad_keeps_fp32_accuracy_at_any_magnitude(scale):
    """The bf16 split has fp32's exponent range: tiny (or huge) gradients lose nothing, and the result
    is as close to the float64 product as an fp32 GEMM would be (the dropped cross terms are 2^-24)."""
    from mlgnn import _lib
    N, M, K = 30000, 256, 128
    gen = torch.Generator().manual_seed(3)
    g = torch.randn(N, M, generator=gen) * scale
    g[::7] *= 1e-3                                         # rows of very different magnitude
    x = torch.randn(N, K, generator=gen)
    ref = g.double().t() @ x.double()
    ref_b = g.double().sum(0)
    mag = (g.double().abs().t() @ x.double().abs())        # size of the terms each output sums
    dev = "cuda:0"
    gd, xd = g.to(dev), x.to(dev)
    n = int(_lib.lib.mlgnn_linear_wgrad_workspace_floats(N, M, K, 0))
    ws = torch.empty(n, device=dev)
    out = torch.empty(M * K + M, device=dev)
    rc = _lib.lib.mlgnn_linear_wgrad(gd.data_ptr(), xd.data_ptr(), None, None, None, None, out.data_ptr(), ws.data_ptr(), n, N, M, K, 0,
                                     torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    got = out[:M * K].view(M, K).cpu().double()
    err = ((got - ref).abs() / mag).max().item()
    assert err < 1e-6, err                                 # 2^-20: elementwise, relative to the summed magnitudes
    lib = (gd.t() @ xd).cpu().double()                     # the library's fp32 GEMM, for scale
    lib_err = ((lib - ref).abs() / mag).max().item()
    assert err < 4 * lib_err + 1e-7, (err, lib_err)
    assert_close(out[M * K:].cpu().double() / scale, ref_b / scale, 1e-5, "bias grad")
    # bitwise reproducible (fixed summation order)
    out2 = torch.empty_like(out)
    _lib.lib.mlgnn_linear_wgrad(gd.data_ptr(), xd.data_ptr(), None, None, None, None, out2.data_ptr(), ws.data_ptr(), n, N, M, K, 0,
                                torch.cuda.current_stream().cuda_stream)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("scale", [1.0, 1e-12, 1e9])
@pytest.mark.parametrize("slack", [1.0, 37.0])
def test_wgrad_scaled_fp16_split_accuracy(scale, slack):
    """With max |grad_out| and max |x| supplied the kernel takes the scaled two-way fp16 split (3 MFMAs per product):
    still fp32-GEMM-level error at any magnitude, also when the supplied bounds are loose (`slack` x the true maxima),
    for rows a thousand times smaller than the largest, and bitwise reproducible."""
    from mlgnn import _lib
    N, M, K = 30000, 128, 256
    gen = torch.Generator().manual_seed(4)
    g = torch.randn(N, M, generator=gen) * scale
    g[::7] *= 1e-3
    x = torch.randn(N, K, generator=gen)
    x[::11] *= 1e-2
    ref = g.double().t() @ x.double()
    mag = (g.double().abs().t() @ x.double().abs())
    dev = "cuda:0"
    gd, xd = g.to(dev), x.to(dev)
    gmax, xmax = gd.abs().amax(1) * slack, xd.abs().amax(1) * slack              # row maxima (x a loose factor)
    n = int(_lib.lib.mlgnn_linear_wgrad_workspace_floats(N, M, K, 0))
    ws = torch.empty(n, device=dev)
    out = torch.empty(M * K + M, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    rc = _lib.lib.mlgnn_linear_wgrad(gd.data_ptr(), xd.data_ptr(), None, None, gmax.data_ptr(), xmax.data_ptr(), out.data_ptr(),
                                     ws.data_ptr(), n, N, M, K, 0, st)
    assert rc == 0
    got = out[:M * K].view(M, K).cpu().double()
    err = ((got - ref).abs() / mag).max().item()
    lib = (gd.t() @ xd).cpu().double()
    lib_err = ((lib - ref).abs() / mag).max().item()
    assert err < 1e-6, err
    assert err < 4 * lib_err + 1e-7, (err, lib_err)
    assert_close(out[M * K:].cpu().double() / scale, g.double().sum(0) / scale, 1e-5, "bias grad")
    out2 = torch.empty_like(out)
    _lib.lib.mlgnn_linear_wgrad(gd.data_ptr(), xd.data_ptr(), None, None, gmax.data_ptr(), xmax.data_ptr(), out2.data_ptr(),
                                ws.data_ptr(), n, N, M, K, 0, st)
    assert torch.equal(out, out2)
    # the small rows alone: their share of the result keeps its accuracy next to the large rows
    small = torch.zeros(N, dtype=torch.bool)
    small[::7] = True
    ref_s = g[small].double().t() @ x[small].double()
    g2 = torch.where(small[:, None], g, torch.zeros_like(g)).to(dev)
    _lib.lib.mlgnn_linear_wgrad(g2.data_ptr(), xd.data_ptr(), None, None, gmax.data_ptr(), xmax.data_ptr(), out2.data_ptr(),
                                ws.data_ptr(), n, N, M, K, 0, st)
    mag_s = (g[small].double().abs().t() @ x[small].double().abs())
    err_s = ((out2[:M * K].view(M, K).cpu().double() - ref_s).abs() / mag_s).max().item()
    assert err_s < (2e-5 if slack > 1 else 2e-6), err_s    # 2^-10 of the maximum (x a loose bound): bits run out gradually


def test_unsupported_shapes_use_library_gemm():
    from mlgnn import _lib
    from mlgnn.dense import linear
    assert _lib.lib.mlgnn_linear_wgrad_workspace_floats(10000, 256, 256, 0) < 0      # 64 tiles > 32
    x = torch.randn(9000, 256, device="cuda:0", requires_grad=True)
    w = torch.randn(256, 256, device="cuda:0", requires_grad=True)
    linear(x, w).sum().backward()
    assert_close(w.grad, x.detach().sum(0)[None, :].expand(256, 256), 1e-4)


@pytest.mark.parametrize("N,M,K", [(9001, 37, 128), (10000, 64, 64), (8195, 128, 10), (12345, 256, 128), (8192, 32, 32),
                                   (20011, 200, 40)])
@pytest.mark.parametrize("act", [False, True])
def test_wgrad_scaled_split_every_layout(N, M, K, act):
    """The scaled fp16 split through every kernel layout the planner picks (4- and 8-wave workgroups, padded tiles, the
    masked remainder launch, the affine + ReLU operand prologue) against fp64."""
    from mlgnn.dense import _wgrad
    gen = torch.Generator().manual_seed(N + M)
    go = torch.randn(N, M, generator=gen)
    x = torch.randn(N, K, generator=gen)
    gamma, beta = torch.rand(K, generator=gen) + 0.5, torch.randn(K, generator=gen) * 0.3
    xa = torch.relu(x.double() * gamma.double() + beta.double()) if act else x.double()
    ref_w, ref_b = go.double().t() @ xa, go.double().sum(0)
    mag = go.double().abs().t() @ xa.abs() + 1e-30
    god, xd = go.cuda(), x.cuda()
    kw = dict(x_gamma=gamma.cuda(), x_beta=beta.cuda()) if act else {}
    gmax = god.abs().amax(1)
    xmax = xa.abs().amax(1).float().cuda()
    w, b = _wgrad(god, xd, go_max=gmax, x_max=xmax, **kw)
    w0, b0 = _wgrad(god, xd, **kw)                                  # the exact three-way split, for scale
    err = ((w.cpu().double() - ref_w).abs() / mag).max().item()
    err0 = ((w0.cpu().double() - ref_w).abs() / mag).max().item()
    assert err < 1e-6 and err < 8 * err0 + 2e-7, (err, err0)
    assert_close(b.cpu().double(), ref_b, 1e-5, "bias grad")


def test_wgrad_scaled_split_full_size_against_exact_split():
    """BASELINE configs[1] size (640 000 rows, 128 x 256, the ReLU-affine operand prologue): the scaled fp16 split against
    the exact three-way bf16 split of the same kernel, relative to the magnitudes each output sums; bias gradients equal."""
    from mlgnn.dense import _wgrad
    gen = torch.Generator(device="cuda").manual_seed(8)
    N, M, K = 640000, 128, 256
    go = torch.randn(N, M, device="cuda", generator=gen)
    go[::9] *= 1e-3
    xhat = torch.nn.functional.layer_norm(torch.randn(N, K, device="cuda", generator=gen), (K,))
    gamma, beta = torch.rand(K, device="cuda", generator=gen) + 0.5, torch.randn(K, device="cuda", generator=gen) * 0.2
    act = torch.relu(xhat * gamma + beta)
    w, b = _wgrad(go, xhat, gamma, beta, go_max=go.abs().amax(1), x_max=act.amax(1))
    w0, b0 = _wgrad(go, xhat, gamma, beta)
    mag = go.abs().t() @ act
    err = ((w - w0).abs() / mag).max().item()
    assert err < 1e-6, err
    assert torch.equal(b, b0)
