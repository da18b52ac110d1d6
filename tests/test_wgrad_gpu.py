"""Split-row fp32-MFMA weight/bias gradient kernel vs torch autograd on the CPU."""
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


def test_unsupported_shapes_use_library_gemm():
    from mlgnn import _lib
    from mlgnn.dense import linear
    assert _lib.lib.mlgnn_linear_wgrad_workspace_floats(10000, 256, 256) < 0      # 64 tiles > 32
    x = torch.randn(9000, 256, device="cuda:0", requires_grad=True)
    w = torch.randn(256, 256, device="cuda:0", requires_grad=True)
    linear(x, w).sum().backward()
    assert_close(w.grad, x.detach().sum(0)[None, :].expand(256, 256), 1e-4)
