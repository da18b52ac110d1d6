"""Linear with a handful of input columns over tall rows (csrc/sage.hip narrow_linear: the node encoder Linear(3, hidden)
of models/deepergcn.py:199-210) against fp64."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("N", [8192, 10007, 640000])
@pytest.mark.parametrize("R,J", [(3, 128), (1, 32), (4, 64), (8, 256), (7, 128)])
@pytest.mark.parametrize("bias", [True, False])
def test_narrow_linear_matches_fp64(N, R, J, bias):
    from mlgnn import dense as D
    g = torch.Generator(device=DEV).manual_seed(N + 31 * R + J)
    x = torch.randn(N, R, device=DEV, generator=g)
    lin = torch.nn.Linear(R, J, bias=bias).to(DEV)
    y = D.linear(x, lin.weight, lin.bias)
    assert type(y.grad_fn).__name__.startswith("_NarrowLinear")          # the stream kernels ran, not a library GEMM
    ref = torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double() if bias else None)
    assert float((y.double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    cot = torch.randn(N, J, device=DEV, generator=g)
    (y * cot).sum().backward()
    gw = cot.double().t() @ x.double()
    # (a sum of N products: fp32 accumulation in a fixed tree, error ~ sqrt(N) ulps of the summed magnitudes)
    assert float((lin.weight.grad.double() - gw).abs().max()) <= 2e-6 * float((cot.double().abs().t() @ x.double().abs()).max())
    if bias:
        assert float((lin.bias.grad.double() - cot.double().sum(0)).abs().max()) <= 2e-6 * float(cot.double().abs().sum(0).max())
    # bitwise reproducible
    lin.weight.grad = None
    y2 = D.linear(x, lin.weight, lin.bias)
    (y2 * cot).sum().backward()
    assert torch.equal(y, y2)


def test_narrow_linear_input_gradient():
    from mlgnn import dense as D
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(9000, 3, device=DEV, generator=g, requires_grad=True)
    w = torch.randn(128, 3, device=DEV, generator=g, requires_grad=True)
    cot = torch.randn(9000, 128, device=DEV, generator=g)
    (D.linear(x, w) * cot).sum().backward()
    assert torch.allclose(x.grad, cot @ w.detach(), rtol=1e-5, atol=1e-5)
