"""Large bf16 GEMM (csrc/gemm_nt.hip) against torch: the operands are bf16, so every product is exact in fp32 and the
only difference is the fp32 summation order: |err| <= 2e-6 * sum_k |a_k b_k| (K <= 4096 terms)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(segments):
    acc = None
    for a, b in segments:
        t = a.double() @ b.double().t()
        acc = t if acc is None else acc + t
    return acc


def _bound(segments):
    acc = None
    for a, b in segments:
        t = a.double().abs() @ b.double().abs().t()
        acc = t if acc is None else acc + t
    return acc


@pytest.mark.parametrize("M,N,K,splits", [(128, 128, 64, 1), (256, 384, 192, 1), (1024, 256, 4096, 1), (512, 512, 1024, 4),
                                          (128, 256, 128, 2), (384, 128, 320, 5), (128, 128, 64, 1)])
def test_gemm_nt_matches_fp64(M, N, K, splits):
    from mlgnn.gemm import gemm_bf16_nt
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).cuda().bfloat16()
    b = (torch.randn(N, K, generator=g) + 0.3).cuda().bfloat16()      # asymmetric, non-zero mean
    ref, bound = _ref([(a, b)]), _bound([(a, b)])
    if splits > 1:
        got = gemm_bf16_nt([(a, b)], splits=splits)["slab"].double().sum(0)
        assert ((got - ref).abs() <= 2e-6 * bound + 1e-30).all()
        return
    o = gemm_bf16_nt([(a, b)], out_dtype=torch.float32, want_ct=True, dot=a[:, :N].contiguous() if K >= N else None)
    assert ((o["c"].double() - ref).abs() <= 2e-6 * bound).all()
    # the transposed bf16 copy is the correctly rounded fp32 result
    assert torch.equal(o["ct"], o["c"].t().bfloat16())
    if "dot" in o:
        d = a[:, :N].double()
        want = float((d * ref).sum())
        assert abs(float(o["dot"]) - want) <= 1e-5 * float((d.abs() * bound).sum())
    ob = gemm_bf16_nt([(a, b)])["c"]
    assert torch.equal(ob, o["c"].bfloat16())


def test_gemm_nt_segments_strided_aux():
    """Several (A_s, B_s) terms summed by one launch; operands are column slices of wider buffers; aux epilogue."""
    from mlgnn.gemm import gemm_bf16_nt
    g = torch.Generator().manual_seed(5)
    M, N = 256, 128
    wide_a = torch.randn(M, 512, generator=g).cuda().bfloat16()
    wide_b = torch.randn(N, 512, generator=g).cuda().bfloat16()
    a2 = torch.randn(M, 64, generator=g).cuda().bfloat16()
    b2 = torch.randn(N, 64, generator=g).cuda().bfloat16()
    segs = [(wide_a[:, 64:192], wide_b[:, 128:256]), (a2, b2), (wide_a[:, 256:512], wide_b[:, 0:256])]
    aux = torch.randn(M, N, generator=g).cuda()
    o = gemm_bf16_nt(segs, out_dtype=torch.float32, aux=aux, alpha=-0.75)
    ref = _ref(segs) - 0.75 * aux.double()
    assert ((o["c"].double() - ref).abs() <= 2e-6 * (_bound(segs) + aux.double().abs())).all()
    for sp in (2, 7):
        got = gemm_bf16_nt(segs, splits=sp)["slab"].double().sum(0)
        assert ((got - _ref(segs)).abs() <= 2e-6 * _bound(segs)).all()


def test_gemm_nt_identity_asymmetric():
    """A = I against an asymmetric B: catches a transposed C write or a wrong fragment map (exact integers)."""
    from mlgnn.gemm import gemm_bf16_nt
    n = 256
    a = torch.eye(n).cuda().bfloat16()
    b = (torch.arange(n).view(n, 1) * 2 % 97 + torch.arange(n).view(1, n) % 13).float().cuda().bfloat16()   # b[j][k]
    o = gemm_bf16_nt([(a, b)], out_dtype=torch.float32, want_ct=True)
    assert torch.equal(o["c"], b.float().t())
    assert torch.equal(o["ct"].float(), b.float())


def test_gemm_nt_argument_errors():
    from mlgnn import _lib
    from mlgnn.gemm import gemm_bf16_nt
    a = torch.zeros(100, 64, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(_lib.MlgnnError):
        gemm_bf16_nt([(a, a)])
    a = torch.zeros(128, 96, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(_lib.MlgnnError):
        gemm_bf16_nt([(a, a)])
