"""Large bf16 GEMM on the matrix cores (``mlgnn_gemm_bf16_nt``): ``C = sum_s A_s B_s^T`` with fp32 accumulation.

The dense products of the DiffPool contraction at BASELINE configs[4] size and their gradients
(reference: ``dense_diff_pool`` as called from ``models/diff_pooling.py:59-65``)."""
import ctypes

import torch

from . import _lib

TILE = 128
BK = 64


def _dtype_id(t):
    return 1 if t.dtype == torch.bfloat16 else 0


def gemm_supported(M, N, ks):
    return M > 0 and N > 0 and M % TILE == 0 and N % TILE == 0 and 1 <= len(ks) <= 4 and all(k > 0 and k % BK == 0 for k in ks)


def gemm_bf16_nt(segments, splits=1, out_dtype=torch.bfloat16, want_c=True, want_ct=False, aux=None, alpha=0.0,
                 dot=None, slab=None):
    """``segments``: list of ``(A [M,K_s], B [N,K_s])`` bf16 (rows may be strided views, unit stride along K).
    Returns a dict with the requested ones of ``c`` [M,N], ``ct`` [N,M] bf16, ``dot`` (scalar: ``sum(dot * C)``),
    ``slab`` [splits,M,N] fp32 (``splits > 1``: partial results, nothing else is produced)."""
    A0, B0 = segments[0]
    M, N = A0.shape[0], B0.shape[0]
    dev = A0.device
    n = len(segments)
    pa, pb = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
    la, lb, ks = (ctypes.c_int64 * n)(), (ctypes.c_int64 * n)(), (ctypes.c_int64 * n)()
    for i, (a, b) in enumerate(segments):
        if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16 or not a.is_cuda:
            raise TypeError("gemm_bf16_nt wants bf16 device operands")
        if a.shape[0] != M or b.shape[0] != N or a.shape[1] != b.shape[1] or a.stride(1) != 1 or b.stride(1) != 1:
            raise ValueError("operand shapes / strides")
        pa[i], pb[i] = a.data_ptr(), b.data_ptr()
        la[i], lb[i], ks[i] = a.stride(0), b.stride(0), a.shape[1]
    out = {}
    c = ct = partial = None
    if splits > 1:
        if slab is None:
            slab = torch.empty((splits, M, N), dtype=torch.float32, device=dev)
        out["slab"] = slab
    else:
        if want_c:
            c = out["c"] = torch.empty((M, N), dtype=out_dtype, device=dev)
        if want_ct:
            ct = out["ct"] = torch.empty((N, M), dtype=torch.bfloat16, device=dev)
        if dot is not None:
            partial = torch.empty(int(_lib.lib.mlgnn_gemm_bf16_nt_workgroups(M, N, 1)), dtype=torch.float32, device=dev)
    rc = _lib.lib.mlgnn_gemm_bf16_nt(
        pa, pb, la, lb, ks, n, M, N, splits, _lib.ptr(slab if splits > 1 else None),
        _lib.ptr(c), N, _dtype_id(c) if c is not None else 1, _lib.ptr(ct), M,
        _lib.ptr(aux), aux.stride(0) if aux is not None else 0, _dtype_id(aux) if aux is not None else 0, float(alpha),
        _lib.ptr(dot), dot.stride(0) if dot is not None else 0, _lib.ptr(partial),
        torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "mlgnn_gemm_bf16_nt")
    if partial is not None:
        out["dot"] = partial.sum()
    return out
