"""fp32 nn.Linear at widths past the fp32 tall kernels (hidden 512 at BASELINE configs[4]'s d = 256):
``mlgnn_linear_f32x3_fwd`` / ``_bwd`` -- three-term bf16 products -- against fp64 (torch_nn.py:54-75)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,R,J,bias", [(20000, 256, 512, True), (20001, 512, 256, True), (9000, 128, 640, False),
                                        (8192, 512, 512, True)])
def test_wide_fp32_linear_matches_fp64(N, R, J, bias):
    """Output, input gradient and weight gradient within 1e-4 of fp64, elementwise against each product's absolute-value
    bound (the three-term split leaves 2^-17 per operand); ragged row counts (padding rows are zeros inside the
    workspace); the op is the one ``mlgnn.dense.linear`` picks for these shapes."""
    from mlgnn import dense
    g = torch.Generator().manual_seed(N + R)
    dev = "cuda:0"
    x = torch.randn(N, R, generator=g).to(dev)
    w = (torch.randn(J, R, generator=g) * R ** -0.5).to(dev)
    b = torch.randn(J, generator=g).to(dev) if bias else None
    cot = torch.randn(N, J, generator=g).to(dev)
    assert dense._lib.lib.mlgnn_linear_f32x3_supported(N, R, J) == 1
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    bd = b.double().requires_grad_(True) if bias else None
    ref = torch.nn.functional.linear(xd, wd, bd)
    (ref * cot.double()).sum().backward()
    xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    bc = b.clone().requires_grad_(True) if bias else None
    before = torch.cuda.memory_allocated()
    y = dense.linear(xc, wc, bc)
    assert isinstance(y.grad_fn, dense._WideLinearF32._backward_cls) or "WideLinearF32" in type(y.grad_fn).__name__
    assert y.shape == (N, J) and y.is_contiguous()
    bound = x.double().abs() @ w.double().abs().t() + (b.double().abs() if bias else 0.0)
    assert ((y.double() - ref.detach()).abs() <= 1e-4 * bound).all()
    (y * cot).sum().backward()
    gx_bound = cot.double().abs() @ w.double().abs()
    assert ((xc.grad.double() - xd.grad).abs() <= 1e-4 * gx_bound).all()
    gw_bound = cot.double().abs().t() @ x.double().abs()
    assert ((wc.grad.double() - wd.grad).abs() <= 1e-4 * gw_bound).all()
    if bias:
        assert ((bc.grad.double() - bd.grad).abs() <= 1e-5 * cot.double().abs().sum(0)).all()
    del before


def test_wide_fp32_linear_is_repeatable_and_refuses_bad_arguments():
    from mlgnn import _lib, dense
    N, R, J = 8192, 256, 512
    x, w = torch.randn(N, R, device="cuda"), torch.randn(J, R, device="cuda")
    outs = []
    for _ in range(2):
        xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = dense.linear(xc, wc, None)
        y.square().sum().backward()
        outs.append((y.detach(), xc.grad, wc.grad))
    for u, v in zip(*outs):
        assert torch.equal(u, v)
    L = _lib.lib
    assert L.mlgnn_linear_f32x3_supported(N, 200, J) == 0 and L.mlgnn_linear_f32x3_supported(0, R, J) == 0
    assert L.mlgnn_linear_f32x3_fwd_workspace_bytes(N, 200, J) == -2
    assert L.mlgnn_linear_f32x3_fwd(None, None, None, None, None, 0, N, R, J, None) == -1
    need = L.mlgnn_linear_f32x3_fwd_workspace_bytes(N, R, J)
    y = torch.empty(N, J, device="cuda")
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    assert L.mlgnn_linear_f32x3_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), ws.data_ptr(), need - 1, N, R, J, None) == -5
