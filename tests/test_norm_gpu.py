"""Fused LayerNorm(+ReLU) kernels vs torch.nn.functional on the CPU (fp32, 1e-4)."""
import pytest
import torch
import torch.nn.functional as F

from _util import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,d", [(1, 4), (7, 8), (1000, 32), (513, 64), (2049, 100), (4097, 128), (3000, 256),
                                    (70000, 128)])
@pytest.mark.parametrize("relu", [False, True])
def test_layer_norm_act(rows, d, relu):
    from mlgnn.norm import layer_norm_act
    gen = torch.Generator().manual_seed(rows + d)
    x = (torch.randn(rows, d, generator=gen) * 2 + 0.5).requires_grad_(True)
    w = (torch.rand(d, generator=gen) + 0.5).requires_grad_(True)
    b = (torch.randn(d, generator=gen) * 0.3).requires_grad_(True)
    cot = torch.randn(rows, d, generator=gen)
    ref = F.layer_norm(x, (d,), w, b, 1e-5)
    ref = F.relu(ref) if relu else ref
    gr = torch.autograd.grad((ref * cot).sum(), [x, w, b])
    dev = "cuda:0"
    xd, wd, bd = (t.detach().to(dev).requires_grad_(True) for t in (x, w, b))
    out = layer_norm_act(xd, wd, bd, 1e-5, relu)
    assert_close(out, ref, 1e-4, "ln fwd")
    got = torch.autograd.grad((out * cot.to(dev)).sum(), [xd, wd, bd])
    for name, g, r in zip(("x", "gamma", "beta"), got, gr):
        assert_close(g, r, 1e-4, "ln grad " + name)


def test_unsupported_width_uses_aten_on_device():
    from mlgnn.norm import layer_norm_act
    x = torch.randn(10, 258, device="cuda:0")
    w, b = torch.ones(258, device="cuda:0"), torch.zeros(258, device="cuda:0")
    assert_close(layer_norm_act(x, w, b, 1e-5, True), F.relu(F.layer_norm(x, (258,), w, b)), 1e-6)


@pytest.mark.parametrize("rows,d", [(1, 4), (1000, 32), (777, 100), (5000, 128), (3000, 256)])
def test_msg_norm_add(rows, d):
    from mlgnn.norm import msg_norm_add
    gen = torch.Generator().manual_seed(rows * 3 + d)
    x = torch.randn(rows, d, generator=gen, requires_grad=True)
    m = (torch.rand(rows, d, generator=gen) * 3).requires_grad_(True)
    with torch.no_grad():
        m[0] = 0.0                                     # zero message row: F.normalize clamps the norm
        if rows > 2:
            x[2] = 0.0                                 # zero feature row: ||x|| has a zero sub-gradient
    s = torch.tensor([0.7], requires_grad=True)
    cot = torch.randn(rows, d, generator=gen)
    ref = x + F.normalize(m, p=2.0, dim=1) * x.norm(p=2, dim=1, keepdim=True) * s
    gr = torch.autograd.grad((ref * cot).sum(), [x, m, s])
    dev = "cuda:0"
    xd, md, sd = (t.detach().to(dev).requires_grad_(True) for t in (x, m, s))
    out = msg_norm_add(xd, md, sd)
    assert_close(out, ref, 1e-4, "msgnorm fwd")
    got = torch.autograd.grad((out * cot.to(dev)).sum(), [xd, md, sd])
    for name, g, r in zip(("x", "m", "scale"), got, gr):
        assert_close(g, r, 1e-4, "msgnorm grad " + name)
