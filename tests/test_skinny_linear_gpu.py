"""Linear over a handful of rows with a very long input (csrc/skinny.hip: the first layer of MultilevelGNN's head,
models/multilevel_gnn.py:121-127 -- Linear(84 096, 512) on 64 samples at config/kirc.yaml) against fp64."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("M,J,K", [(64, 512, 84096), (2, 512, 84096), (32, 256, 8192), (7, 96, 8196), (64, 100, 12300)])
@pytest.mark.parametrize("bias", [True, False])
def test_skinny_linear_matches_fp64(M, J, K, bias):
    from mlgnn import dense as D
    g = torch.Generator(device=DEV).manual_seed(M + J + K)
    x = torch.randn(M, K, device=DEV, generator=g, requires_grad=True)
    lin = torch.nn.Linear(K, J, bias=bias).to(DEV)
    y = D.linear(x, lin.weight, lin.bias)
    assert type(y.grad_fn).__name__.startswith("_SkinnyLinear")
    ref = torch.nn.functional.linear(x.detach().double(), lin.weight.double(), lin.bias.double() if bias else None)
    # fp32 FMA chains over K terms: a few ulps of the summed magnitudes
    mag = x.detach().double().abs() @ lin.weight.double().abs().t()
    assert bool(((y.double() - ref).abs() <= 2e-6 * mag + 1e-30).all())
    cot = torch.randn(M, J, device=DEV, generator=g)
    (y * cot).sum().backward()
    gx = cot.double() @ lin.weight.double()
    assert bool(((x.grad.double() - gx).abs() <= 2e-6 * (cot.double().abs() @ lin.weight.double().abs()) + 1e-30).all())
    gw = cot.double().t() @ x.detach().double()
    assert bool(((lin.weight.grad.double() - gw).abs() <= 2e-6 * (cot.double().abs().t() @ x.detach().double().abs()) + 1e-30).all())
    if bias:
        assert torch.allclose(lin.bias.grad.double(), cot.double().sum(0), rtol=1e-5, atol=1e-5 * float(cot.abs().sum(0).max()))
    y2 = D.linear(x, lin.weight, lin.bias)
    assert torch.equal(y, y2)                                   # fixed summation order


def test_weight_gradient_lands_in_the_flat_bucket_without_a_copy():
    """Under FlatAdam / FlatGradBucket the 172 MB weight gradient is written into the parameter's slot of the flat buffer
    by the kernel itself: after backward ``p.grad`` aliases the slot (nothing for ``collect()`` to move), the values are
    the gradient, and accumulating a SECOND backward into a kept gradient still adds (memory of its own)."""
    from mlgnn import dense as D
    from mlgnn.dist import FlatGradBucket
    g = torch.Generator(device=DEV).manual_seed(3)
    lin = torch.nn.Linear(16384, 64).to(DEV)
    head = torch.nn.Linear(64, 2).to(DEV)
    mods = torch.nn.ModuleList([lin, head])
    bucket = FlatGradBucket(mods)
    x = torch.randn(8, 16384, device=DEV, generator=g)

    def loss():
        return head(torch.relu(D.linear(x, lin.weight, lin.bias))).square().mean()

    bucket.release()
    loss().backward()
    slot = lin.weight._mlgnn_grad_slot
    assert lin.weight.grad.data_ptr() == slot.data_ptr()          # adopted, not copied
    bucket.collect()
    assert bucket.check_views()
    ref = torch.autograd.grad(head(torch.relu(torch.nn.functional.linear(x, lin.weight, lin.bias))).square().mean(),
                              [lin.weight])[0]
    assert torch.allclose(lin.weight.grad, ref, rtol=1e-4, atol=1e-7)
    # accumulate on top of the kept gradient (the zero()-style flow): the sum, not an aliasing accident
    before = lin.weight.grad.clone()
    loss().backward()
    assert torch.allclose(lin.weight.grad, 2 * before, rtol=1e-5, atol=1e-7)
