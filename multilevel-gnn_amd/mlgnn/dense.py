"""Dense pooled-graph operators of DiffPool (reference: PyG 2.2.0 ``DenseSAGEConv`` and
``dense_diff_pool`` as called from ``models/diff_pooling.py:24-32,36,45,64``)."""
import os

import torch
import torch.nn.functional as F

from . import _lib
from .ops import _DTYPE_IDS, f32_cached, row_max_of

DIFFPOOL_EPS = 1e-15
WGRAD_MIN_ROWS = 8192          # below this the library's TN GEMM is not the bottleneck
ROW_MAX_STATS = {"given": 0, "computed": 0}      # tall GEMMs whose operand came with / without its row maxima


def _aligned(t):
    """Contiguous and on a 16-byte boundary (the kernels read operands with 16-byte loads): a view at an odd offset of
    somebody's flat buffer is copied once (mlgnn.optim.FlatAdam aligns its slots, so its views never are)."""
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.contiguous_format)


def tall_matmul_nt(a, bt, bias=None, residual=None, row_max=None, ln=None, bt_transposed=False):
    """``a [N,R] @ bt[J,R]^T (+ bias) (+ residual [N,J])`` through the scaled split-precision fp16-MFMA kernel
    (``csrc/tallgemm.hip``).  The caller checks :func:`tall_matmul_supported` first.  ``row_max`` [N]: ``max |a[i]|``
    when the producer of ``a`` supplied it (see :func:`mlgnn.ops.tag_row_max`).
    ``ln = ("out", gamma, beta, eps)``: returns ``(xhat, rstd, row_max_y)`` -- the layer-normalised result without
    the affine map, its 1/sigma and ``max relu(gamma xhat + beta)`` per row.  ``ln = ("in", gamma, beta)``: ``a`` is
    such an ``xhat``; ``relu(gamma a + beta)`` is applied as it is loaded."""
    N, R = a.shape
    if bt_transposed and a.dtype != torch.float32:           # (the bf16 kernel packs a [J, R] operand)
        bt, bt_transposed = bt.t(), False
    J = bt.shape[1] if bt_transposed else bt.shape[0]        # bt_transposed: bt is [R, J], read with swapped indices
    a, bt = _aligned(a), _aligned(bt)
    dt = _DTYPE_IDS[a.dtype]
    out = torch.empty((N, J), dtype=a.dtype, device=a.device)
    nbytes = int(_lib.lib.mlgnn_tallgemm_workspace_bytes(R, J, dt))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
    if residual is not None:
        residual = residual.contiguous()
    if bias is not None:
        bias = f32_cached(bias)                          # [J]; the kernels add it in fp32
    if a.dtype == torch.float32:
        ROW_MAX_STATS["given" if row_max is not None else "computed"] += 1
    mode, gamma, beta, eps, rstd, rmax = 0, None, None, 0.0, None, None
    if ln is not None:
        mode = 1 if ln[0] == "out" else 2
        gamma, beta = ln[1].contiguous(), ln[2].contiguous()
        if mode == 1:
            eps = float(ln[3])
            rstd = torch.empty(N, dtype=torch.float32, device=a.device)
            rmax = torch.empty(N, dtype=torch.float32, device=a.device)
    rc = _lib.lib.mlgnn_tallgemm_nt(a.data_ptr(), bt.data_ptr(), int(bt_transposed), _lib.ptr(bias), _lib.ptr(residual),
                                    _lib.ptr(row_max),
                                    mode, _lib.ptr(gamma), _lib.ptr(beta), eps, _lib.ptr(rstd), _lib.ptr(rmax),
                                    out.data_ptr(), ws.data_ptr(), nbytes, N, R, J, dt,
                                    torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "mlgnn_tallgemm_nt")
    return (out, rstd, rmax) if mode == 1 else out


def tall_matmul_lnin_postln(xhat, w, bias, residual, row_max, gamma, beta, post):
    """Second Linear of the fused MLP (``ln = ("in", ...)`` of :func:`tall_matmul_nt`) with a further LayerNorm of the
    result rows in the epilogue: ``post = (gamma2, beta2, eps2, relu)`` -> ``(out, y, mean2, rstd2)`` with
    ``y = relu?(LayerNorm(out))`` (csrc/tallgemm.hip POST)."""
    N, R = xhat.shape
    J = w.shape[0]
    xhat, w = _aligned(xhat), _aligned(w)
    f32 = dict(dtype=torch.float32, device=xhat.device)
    out, y = torch.empty((N, J), **f32), torch.empty((N, J), **f32)
    mean, rstd = torch.empty(N, **f32), torch.empty(N, **f32)
    nbytes = int(_lib.lib.mlgnn_tallgemm_workspace_bytes(R, J, 0))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=xhat.device)
    if residual is not None:
        residual = residual.contiguous()
    if bias is not None:
        bias = f32_cached(bias)
    ROW_MAX_STATS["given" if row_max is not None else "computed"] += 1
    g2, b2, eps2, relu = post
    rc = _lib.lib.mlgnn_tallgemm_lnin_postln(xhat.data_ptr(), w.data_ptr(), _lib.ptr(bias), _lib.ptr(residual),
                                             _lib.ptr(row_max), gamma.contiguous().data_ptr(),
                                             beta.contiguous().data_ptr(), g2.contiguous().data_ptr(),
                                             b2.contiguous().data_ptr(), float(eps2), int(bool(relu)), out.data_ptr(),
                                             y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), nbytes,
                                             N, R, J, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "mlgnn_tallgemm_lnin_postln")
    return out, y, mean, rstd


def tall_matmul_nt_shift(a, w, row_max, lse, rowptr):
    """Input gradient ``a [N,R] @ w [R,J]`` (``w``: the Linear's own weight) of a Linear that reads the output of a softmax
    aggregation, with that aggregation's rescaled cotangent from the same epilogue (csrc/tallgemm.hip SHIFT):
    ``-> (gx, gx * 2^(-lse), flag)``."""
    N, R = a.shape
    J = w.shape[1]
    a, w = _aligned(a), _aligned(w)
    gx = torch.empty((N, J), dtype=torch.float32, device=a.device)
    gt = torch.empty_like(gx)
    flag = torch.empty(4, dtype=torch.int32, device=a.device)
    nbytes = int(_lib.lib.mlgnn_tallgemm_workspace_bytes(R, J, 0))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
    ROW_MAX_STATS["given" if row_max is not None else "computed"] += 1
    rc = _lib.lib.mlgnn_tallgemm_nt_shift(a.data_ptr(), w.data_ptr(), 1, _lib.ptr(row_max), lse.data_ptr(),
                                          rowptr.data_ptr(), gx.data_ptr(), gt.data_ptr(), flag.data_ptr(),
                                          ws.data_ptr(), nbytes, N, R, J, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "mlgnn_tallgemm_nt_shift")
    return gx, gt, flag


def tall_matmul_bf16_shift(go, weight, lse):
    """bf16 storage: input gradient ``go [N,M] @ weight [M,K]`` of a Linear that reads the output of a softmax aggregation,
    with that aggregation's rescaled cotangent from the same epilogue (csrc/tallgemm_bf16.hip SHIFT):
    ``-> (gx, gx * 2^(-lse), flag)``."""
    N, M = go.shape
    K = weight.shape[1]
    go, bt = go.contiguous(), weight.t().contiguous()             # [K, M]: the contraction index contiguous
    gx = torch.empty((N, K), dtype=torch.bfloat16, device=go.device)
    gt = torch.empty_like(gx)
    flag = torch.empty(4, dtype=torch.int32, device=go.device)
    nbytes = int(_lib.lib.mlgnn_tallgemm_workspace_bytes(M, K, 1))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=go.device)
    rc = _lib.lib.mlgnn_tallgemm_bf16_shift(go.data_ptr(), bt.data_ptr(), lse.data_ptr(), gx.data_ptr(), gt.data_ptr(),
                                            flag.data_ptr(), ws.data_ptr(), nbytes, N, M, K,
                                            torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "mlgnn_tallgemm_bf16_shift")
    return gx, gt, flag


def tall_matmul_ln_backward(go, w, xhat, rstd, gamma, beta, row_max=None):
    """``dA = go [N,R] @ w [R,J]`` (``w``: the Linear's own weight) taken through ReLU + LayerNorm backward in the GEMM's
    epilogue (``csrc/tallgemm.hip`` LN = 3): ``-> (grad_h [N,J], grad_gamma, grad_beta, max |grad_h| per row)`` for a
    hidden activation stored normalised (``xhat``, ``rstd``).  The caller checks :func:`tall_matmul_ln_backward_supported`."""
    N, R = go.shape
    J = w.shape[1]
    go, w, xhat = _aligned(go), _aligned(w), _aligned(xhat)
    gh = torch.empty((N, J), dtype=torch.float32, device=go.device)
    ggb = torch.empty((2, J), dtype=torch.float32, device=go.device)
    rmax = torch.empty(N, dtype=torch.float32, device=go.device)
    nbytes = int(_lib.lib.mlgnn_tallgemm_lnbwd_workspace_bytes(R, J))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=go.device)
    ROW_MAX_STATS["given" if row_max is not None else "computed"] += 1
    rc = _lib.lib.mlgnn_tallgemm_lnbwd(go.data_ptr(), w.data_ptr(), 1, _lib.ptr(row_max), xhat.data_ptr(), rstd.data_ptr(),
                                       gamma.contiguous().data_ptr(), beta.contiguous().data_ptr(), gh.data_ptr(),
                                       rmax.data_ptr(), ggb.data_ptr(), ws.data_ptr(), nbytes, N, R, J,
                                       torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "mlgnn_tallgemm_lnbwd")
    return gh, ggb[0], ggb[1], rmax


LB_LN, LB_PLAIN, LB_SHIFT = 0, 1, 2
# bf16: the rescaled cotangent from the input-gradient GEMM's epilogue (csrc/tallgemm_bf16.hip, SHIFT instantiation)
# instead of the streaming pre-pass.  With the lse words requested in the epilogue (32 dependent 8-byte loads per tile)
# it was slower than the pre-pass (65.9 vs 64.7 ms per BASELINE configs[4] step); requested before the k-loop, in an
# instantiation of its own (236 registers, no spills): 58.0 / 58.3 vs 59.2 / 59.5 ms (same box) -- on since round 3.
_BF16_SHIFT = os.environ.get("MLGNN_BF16_SHIFT", "1") == "1"
_ONE_PASS = os.environ.get("MLGNN_ONE_PASS_BWD", "1") == "1"      # (0: the two-kernel backward of each Linear, for A/B runs)
LINEAR_BWD_STATS = {"ln": 0, "shift": 0, "plain": 0}


def linear_backward_supported(N, M, K, epilogue):
    return bool(_lib.lib.mlgnn_linear_bwd_supported(N, M, K, epilogue))


def linear_backward(go, w, x, go_max, x_max, epilogue, rstd=None, gamma=None, beta=None, lse=None,
                    go_max_is_parts=False):
    """Input, weight and bias gradient of ``y = x' W^T + b`` from ONE pass over ``go`` [N,M] and ``x`` [N,K]
    (``csrc/linear_bwd.hip``; ``w`` = the Linear's own weight [M,K]).  ``epilogue``: ``LB_LN`` -- ``x`` is the normalised
    hidden activation, ``x' = relu(gamma x + beta)``, and ``dx`` comes back already taken through ReLU + LayerNorm
    backward; ``LB_PLAIN``; ``LB_SHIFT`` -- also ``dx * 2^(-lse)`` for the softmax aggregation that produced ``x``.
    ``go_max``: row maxima [N] of ``|go|`` or the 256 partial maxima a previous call returned; ``x_max``: row maxima of
    ``|x'|``.  -> ``dict(dx, gw [M,K], gb [M], parts [256], [ggamma, gbeta], [gt, flag])``."""
    N, M = go.shape
    K = x.shape[1]
    go, w, x = _aligned(go), _aligned(w), _aligned(x)
    f32 = dict(dtype=torch.float32, device=go.device)
    dx = torch.empty((N, K), **f32)
    cols = M * K + M + (2 * K if epilogue == LB_LN else 0)
    gwb = torch.empty(cols, **f32)
    parts = torch.empty(256, **f32)
    gt = torch.empty((N, K), **f32) if epilogue == LB_SHIFT else None
    flag = torch.empty(4, dtype=torch.int32, device=go.device) if epilogue == LB_SHIFT else None
    n = int(_lib.lib.mlgnn_linear_bwd_workspace_floats(N, M, K, epilogue))
    _lib.check(min(n, 0), "mlgnn_linear_bwd_workspace_floats")
    ws = torch.empty(n, **f32)
    if gamma is not None:
        gamma, beta = gamma.contiguous(), beta.contiguous()
    rc = _lib.lib.mlgnn_linear_bwd(go.data_ptr(), w.data_ptr(), x.data_ptr(), go_max.data_ptr(), int(go_max_is_parts),
                                   x_max.data_ptr(), epilogue, _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(beta),
                                   _lib.ptr(lse), dx.data_ptr(), _lib.ptr(gt), _lib.ptr(flag), gwb.data_ptr(),
                                   parts.data_ptr(), ws.data_ptr(), n, N, M, K, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "mlgnn_linear_bwd")
    out = dict(dx=dx, gw=gwb[:M * K].view(M, K), gb=gwb[M * K:M * K + M], parts=parts, gt=gt, flag=flag)
    if epilogue == LB_LN:
        out["ggamma"], out["gbeta"] = gwb[M * K + M:M * K + M + K], gwb[M * K + M + K:]
    return out


def tall_matmul_ln_backward_supported(N, R, J):
    return bool(_lib.lib.mlgnn_tallgemm_lnbwd_supported(N, R, J))


def tall_matmul_supported(N, R, J, dtype=torch.float32):
    return dtype in _DTYPE_IDS and bool(_lib.lib.mlgnn_tallgemm_supported(N, R, J, _DTYPE_IDS[dtype]))


_WGRAD_EXACT = os.environ.get("MLGNN_WGRAD_EXACT", "0") == "1"        # always the exact three-way bf16 split


def _wgrad(go, x, x_gamma=None, x_beta=None, go_max=None, x_max=None, out_dtype=None):
    """``(go^T x' [M,K], colsum go [M])`` with ``x' = x`` or ``relu(x_gamma x + x_beta)`` (``csrc/wgrad.hip``).
    ``go_max`` / ``x_max``: ``max |row|`` of ``go`` and of ``x'`` ([N] fp32, the side outputs of the kernels that produced
    them); when both are given (fp32) the kernel uses the scaled two-way fp16 split -- half the MFMAs of the exact
    three-way bf16 split."""
    N, K = x.shape
    M = go.shape[1]
    dt = _DTYPE_IDS[x.dtype]
    n = int(_lib.lib.mlgnn_linear_wgrad_workspace_floats(N, M, K, dt))
    ws = torch.empty(n, dtype=torch.float32, device=x.device)
    out = torch.empty(M * K + M, dtype=torch.float32, device=x.device)          # fp32 for either storage type
    if (x.dtype != torch.float32 or go_max is None or x_max is None or _WGRAD_EXACT
            or go_max.shape[0] != N or x_max.shape[0] != N):
        go_max = x_max = None
    rc = _lib.lib.mlgnn_linear_wgrad(go.data_ptr(), x.data_ptr(), _lib.ptr(x_gamma), _lib.ptr(x_beta), _lib.ptr(go_max),
                                     _lib.ptr(x_max), out.data_ptr(), ws.data_ptr(), n, N, M, K, dt,
                                     torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "mlgnn_linear_wgrad")
    if out_dtype is not None and out_dtype != torch.float32:
        out = out.to(out_dtype)                      # weight and bias gradient of a bf16 model: ONE converting copy
    return out[:M * K].view(M, K), out[M * K:]


class _TallLinear(torch.autograd.Function):
    """``y = x W^T + b`` for a tall ``x [N, K]``: forward and ``dX`` on the tall-matrix MFMA kernels
    (``csrc/tallgemm.hip``: fp32 as split fp16, ``csrc/tallgemm_bf16.hip``: bf16 storage), the fp32
    weight/bias gradient (reduction over the N node rows) on the split-row kernels (``csrc/wgrad.hip``: fp32 as
    3 x bf16; ``csrc/wgrad_bf16.hip``: bf16 operands transposed by the LDS read)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        from .ops import softmax_lse_of
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        # bf16: x is the output of a softmax aggregation -> its backward wants go * 2^(-lse), which the input-gradient
        # GEMM below can write next to go (the fp32 path does this inside the fused MLP)
        ctx.shift_src = softmax_lse_of(x) if (x.dtype == torch.bfloat16 and x.is_contiguous()) else None
        ctx.x_max = row_max_of(x) if x.dtype == torch.float32 else None
        if tall_matmul_supported(x.shape[0], x.shape[1], weight.shape[0], x.dtype):
            # fp32: the kernel holds the residual tile in registers (<= 128 columns); bf16: any width
            fuse = residual is not None and (weight.shape[0] <= 128 or x.dtype == torch.bfloat16)
            out = tall_matmul_nt(x, weight, bias.contiguous() if bias is not None else None, residual if fuse else None,
                                 row_max_of(x))
            return out if (residual is None or fuse) else out + residual
        # addmm on the transposed view picks a faster library kernel than F.linear for these tall
        # shapes (tools/bench_gemm.py: 0.41 vs 0.45 ms at [640k,128] x [128,256])
        out = torch.addmm(bias, x, weight.t()) if bias is not None else torch.mm(x, weight.t())
        return out if residual is None else out + residual

    @staticmethod
    def backward(ctx, go):
        x, weight = ctx.saved_tensors
        go = go.contiguous()
        N, K = x.shape
        M = weight.shape[0]
        gx = None
        if ctx.needs_input_grad[0]:
            src = ctx.shift_src
            if (_BF16_SHIFT and src is not None and go.dtype == torch.bfloat16 and src[0].shape == (N, K)
                    and _lib.lib.mlgnn_tallgemm_bf16_shift_supported(N, M, K)):
                from .ops import tag_shifted
                gx, gt, flag = tall_matmul_bf16_shift(go, weight, src[0])
                tag_shifted(gx, gt, flag, src[0])
            elif tall_matmul_supported(N, M, K, go.dtype):
                gx = tall_matmul_nt(go, weight, row_max=row_max_of(go), bt_transposed=True)     # go [N,M] @ W [M,K]
            else:
                gx = go.matmul(weight)
        gw = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            if x.dtype == torch.float32 or _lib.lib.mlgnn_linear_wgrad_workspace_floats(N, M, K, _DTYPE_IDS[x.dtype]) > 0:
                gw, gb = _wgrad(go, x, go_max=row_max_of(go), x_max=ctx.x_max, out_dtype=x.dtype)
                gb = gb if ctx.has_bias else None
            else:                                    # bf16 widths the transposed-read kernel does not tile
                gw = go.t().mm(x)
                gb = go.sum(0) if ctx.has_bias else None
        # the residual enters by plain addition: its gradient is the output gradient itself (no copy)
        return gx, gw, gb, (go if ctx.needs_input_grad[3] else None)


class _FusedMLP2(torch.autograd.Function):
    """``Linear -> LayerNorm -> ReLU -> Linear (+ residual)`` -- the GENConv MLP (torch_nn.py:54-75,
    torch_vertex.py:35) -- with the hidden activation written ONCE, layer-normalised by the first GEMM's epilogue;
    the affine map + ReLU are applied by its consumers as they load it (second GEMM, its weight gradient), so the
    LayerNorm pass between the two GEMMs does not exist.  The backward is spelled out: weight gradients on the
    split-row kernel, input gradients on the tall GEMM, LayerNorm backward on the stored normalised activation.

    ``post = (gamma2, beta2, eps2, relu)``: the block's NEXT step -- the res+ block's ``norm -> relu`` in front of the
    following conv (deepergcn.py:236-241), or the final norm (:247) -- is computed from the result rows in the second
    GEMM's epilogue; the op then returns ``(out, y)`` with ``y = relu?(LayerNorm(out))`` and its backward starts with
    that LayerNorm's backward, which also adds the gradient arriving on ``out`` (the identity branch of the residual
    block) in the same pass."""

    @staticmethod
    def forward(ctx, x, w1, b1, gamma, beta, w2, b2, residual, eps, post_gamma=None, post_beta=None, post_eps=0.0,
                post_relu=False):
        from .ops import LN_FOLD, PostLN, row_max_of as _rm, softmax_lse_of
        # x is the output of a softmax aggregation: its backward wants go * 2^(-lse), which the input-gradient GEMM
        # below can write next to go
        ctx.shift_src = softmax_lse_of(x) if x.is_contiguous() else None
        # the residual is the `h` of an earlier op's (h, relu(LayerNorm(h))): its gradient can ride that LayerNorm's
        # backward instead of being returned (mlgnn.ops.PostLN)
        ctx.res_tag = getattr(residual, "_mlgnn_post_ln_of", None) if (LN_FOLD and residual is not None) else None
        ctx.post_tag = None
        x = x.contiguous()
        xhat, rstd, rmax = tall_matmul_nt(x, w1, b1, None, _rm(x), ln=("out", gamma, beta, eps))
        fuse = residual is not None and w2.shape[0] <= 128
        ctx.post = post_gamma is not None
        ctx.flags = (b1 is not None, b2 is not None)
        # row maxima of the weight-gradient operands (x; the activated hidden layer): their scales in the backward
        ctx.maxima = (_rm(x), rmax)
        if ctx.post:
            out, y, mean2, rstd2 = tall_matmul_lnin_postln(xhat, w2, b2, residual, rmax, gamma, beta,
                                                           (post_gamma, post_beta, post_eps, post_relu))
            ctx.post_relu = bool(post_relu)
            ctx.post_dtype = post_gamma.dtype
            ctx.set_materialize_grads(False)
            ctx.save_for_backward(x, xhat, rstd, w1, w2, gamma, beta, out, mean2, rstd2, post_gamma, post_beta)
            if LN_FOLD and post_gamma.dtype == torch.float32:
                # (detached alias of `out`: the tag is held by this node, and `out` itself points back to it)
                ctx.post_tag = PostLN(out.detach(), mean2, rstd2, post_gamma.detach(), post_beta.detach(), post_relu)
            return out, y
        out = tall_matmul_nt(xhat, w2, b2, residual if fuse else None, rmax, ln=("in", gamma, beta))
        if residual is not None and not fuse:
            out = out + residual
        ctx.save_for_backward(x, xhat, rstd, w1, w2, gamma, beta)
        return out

    @staticmethod
    def backward(ctx, go, go_y=None):
        from .norm import ln_backward_normalised, ln_backward_saved
        from .ops import row_max_of as _rm
        from .ops import tag_row_max
        gpg = gpb = None
        if ctx.post:
            x, xhat, rstd, w1, w2, gamma, beta, out, mean2, rstd2, pg, pb = ctx.saved_tensors
            # what the consumers of (out, y) left in the side channel (mlgnn.ops.PostLN): the aggregation that read y
            # may already have taken its gradient through this LayerNorm's backward, and the op that added `out` as its
            # residual may have parked that branch's gradient there
            tag, folded, pending = ctx.post_tag, None, None
            if tag is not None:
                folded, tag.folded = tag.folded, None
                if tag.extra is not None and not (folded is not None and tag.extra_used):
                    pending = tag.extra
                tag.extra, tag.extra_used = None, False
            g_out = rmax = None
            if folded is not None:
                g_out, gpg, gpb, rmax = folded
            if go_y is not None:
                # LayerNorm (+ ReLU) backward of y, plus a gradient that arrived on `out` itself, in one pass
                ex, go = go, None
                if ex is None and pending is not None:
                    ex, pending = pending, None
                g2, gpg2, gpb2 = ln_backward_saved(go_y, out, pg, pb, mean2, rstd2, ctx.post_relu, extra=ex)
                if g_out is None:
                    g_out, gpg, gpb = g2, gpg2, gpb2
                else:
                    g_out, gpg, gpb, rmax = g_out + g2, gpg + gpg2, gpb + gpb2, None
            for t in (go, pending):
                if t is not None:
                    g_out, rmax = (t, None) if g_out is None else (g_out + t, None)
            if g_out is None:
                return (None,) * 13
            if rmax is not None:
                tag_row_max(g_out, rmax)
            go = g_out
            gpg, gpb = (gpg.to(ctx.post_dtype), gpb.to(ctx.post_dtype)) if gpg is not None else (None, None)
        else:
            x, xhat, rstd, w1, w2, gamma, beta = ctx.saved_tensors
        has_b1, has_b2 = ctx.flags
        go = go.contiguous()
        x_max, act_max = ctx.maxima
        N = go.shape[0]
        gh_parts = None
        if (_ONE_PASS and _rm(go) is not None and act_max is not None
                and linear_backward_supported(N, w2.shape[0], w2.shape[1], LB_LN)):
            # second Linear: dW2, db2 and dA = go W2 from ONE pass over go and xhat, dA taken through ReLU + LayerNorm
            # backward in the same kernel (csrc/linear_bwd.hip)
            r2 = linear_backward(go, w2, xhat, _rm(go), act_max, LB_LN, rstd=rstd, gamma=gamma, beta=beta)
            gh, gw2, gb2, ggamma, gbeta, gh_parts, gh_max = r2["dx"], r2["gw"], r2["gb"], r2["ggamma"], r2["gbeta"], r2["parts"], None
            LINEAR_BWD_STATS["ln"] += 1
        else:
            gw2, gb2 = _wgrad(go, xhat, gamma.contiguous(), beta.contiguous(), go_max=_rm(go), x_max=act_max)   # go^T relu(gamma xhat + beta)
            if tall_matmul_ln_backward_supported(go.shape[0], go.shape[1], w2.shape[1]):
                # dA = go W2 never reaches memory: ReLU + LayerNorm backward run in the product's epilogue
                gh, ggamma, gbeta, gh_max = tall_matmul_ln_backward(go, w2, xhat, rstd, gamma, beta, _rm(go))
            else:
                gy = tall_matmul_nt(go, w2, row_max=_rm(go), bt_transposed=True)
                gh, ggamma, gbeta, gh_max = ln_backward_normalised(gy, xhat, gamma, beta, rstd, relu=True)
        src = ctx.shift_src if ctx.needs_input_grad[0] else None
        gx = None
        if (_ONE_PASS and x_max is not None and (gh_parts is not None or gh_max is not None)
                and linear_backward_supported(N, w1.shape[0], w1.shape[1], LB_SHIFT)):
            # first Linear: dW1, db1, the input gradient and -- behind a softmax aggregation -- its rescaled cotangent
            # from one pass over gh and x
            from .ops import tag_shifted
            epi = LB_SHIFT if src is not None else LB_PLAIN
            r1 = linear_backward(gh, w1, x, gh_parts if gh_parts is not None else gh_max, x_max, epi,
                                 lse=src[0] if src is not None else None, go_max_is_parts=gh_parts is not None)
            gx, gw1, gb1 = r1["dx"], r1["gw"], r1["gb"]
            if src is not None:
                tag_shifted(gx, r1["gt"], r1["flag"], src[0])
            LINEAR_BWD_STATS["shift" if src is not None else "plain"] += 1
        else:
            gw1, gb1 = _wgrad(gh, x, go_max=gh_max, x_max=x_max)
            if ctx.needs_input_grad[0]:
                if src is not None and _lib.lib.mlgnn_tallgemm_nt_shift_supported(gh.shape[0], gh.shape[1], w1.shape[1]):
                    from .ops import tag_shifted
                    gx, gt, flag = tall_matmul_nt_shift(gh, w1, gh_max, src[0], src[1])
                    tag_shifted(gx, gt, flag, src[0])
                else:
                    gx = tall_matmul_nt(gh, w1, row_max=gh_max, bt_transposed=True)
        g_res = go if ctx.needs_input_grad[7] else None
        if g_res is not None and ctx.res_tag is not None:
            ctx.res_tag.extra, g_res = g_res, None           # added by the LayerNorm backward of the residual's producer
        return (gx, gw1, gb1 if has_b1 else None, ggamma, gbeta, gw2, gb2 if has_b2 else None,
                g_res, None, gpg, gpb, None, None)


def fused_mlp2_supported(x, w1, w2):
    """2-D fp32 CUDA input tall enough for the tall kernels, widths in {64,128,256} with the weight image within LDS."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= WGRAD_MIN_ROWS
            and torch.is_grad_enabled()):
        return False
    k, h, o = x.shape[1], w1.shape[0], w2.shape[0]
    ok = (64, 128, 256)
    return (k in ok and h in ok and o in ok and k * h * 4 <= 128 * 1024 and h * o * 4 <= 128 * 1024 and h <= 256
            and _lib.lib.mlgnn_linear_wgrad_workspace_floats(x.shape[0], h, k, 0) > 0
            and _lib.lib.mlgnn_linear_wgrad_workspace_floats(x.shape[0], o, h, 0) > 0)


def fused_mlp2(x, w1, b1, gamma, beta, eps, w2, b2, residual=None, post_norm=None):
    """``post_norm = (weight, bias, eps, relu)``: also return ``relu?(LayerNorm(out))`` -> ``(out, y)``; the caller checks
    :func:`fused_mlp2_post_supported`."""
    if post_norm is None:
        return _FusedMLP2.apply(x, w1, b1, gamma, beta, w2, b2, residual, float(eps))
    pw, pb, peps, prelu = post_norm
    out, y = _FusedMLP2.apply(x, w1, b1, gamma, beta, w2, b2, residual, float(eps), pw, pb, float(peps), bool(prelu))
    tag = getattr(out.grad_fn, "post_tag", None) if out.grad_fn is not None else None
    if tag is not None:
        from .ops import tag_post_ln
        tag_post_ln(y, out, tag)
    return out, y


def fused_mlp2_post_supported(x, w1, w2, post_weight):
    """The post-LayerNorm epilogue holds whole result rows of 64 or 128 fp32 columns."""
    return (fused_mlp2_supported(x, w1, w2) and post_weight is not None and post_weight.dtype == torch.float32
            and bool(_lib.lib.mlgnn_tallgemm_lnin_postln_supported(x.shape[0], w1.shape[0], w2.shape[0])))


class _WideLinearF32(torch.autograd.Function):
    """``y = x W^T + b`` for a tall fp32 ``x`` whose widths are past the fp32 tall kernels (hidden width 512): forward,
    ``dX`` and ``dW`` as three-term bf16 products on the matrix cores (``mlgnn_linear_f32x3_fwd`` / ``_bwd``,
    csrc/diffpool_large.hip) instead of the library's fp32 GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        N, R = x.shape
        J = weight.shape[0]
        x, weight = x.contiguous(), weight.contiguous()
        Np = int(_lib.lib.mlgnn_linear_f32x3_padded_rows(N))
        y = torch.empty((Np, J), dtype=torch.float32, device=x.device)
        ws = torch.empty(int(_lib.lib.mlgnn_linear_f32x3_fwd_workspace_bytes(N, R, J)), dtype=torch.uint8, device=x.device)
        b = bias.contiguous() if bias is not None else None
        rc = _lib.lib.mlgnn_linear_f32x3_fwd(x.data_ptr(), weight.data_ptr(), _lib.ptr(b), y.data_ptr(), ws.data_ptr(),
                                             ws.numel(), N, R, J, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mlgnn_linear_f32x3_fwd")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y[:N]

    @staticmethod
    def backward(ctx, go):
        x, weight = ctx.saved_tensors
        N, R = x.shape
        J = weight.shape[0]
        go = go.contiguous()
        Np = int(_lib.lib.mlgnn_linear_f32x3_padded_rows(N))
        gx = torch.empty((Np, R), dtype=torch.float32, device=x.device) if ctx.needs_input_grad[0] else None
        gw = torch.empty((J, R), dtype=torch.float32, device=x.device)
        ws = torch.empty(int(_lib.lib.mlgnn_linear_f32x3_bwd_workspace_bytes(N, R, J)), dtype=torch.uint8, device=x.device)
        gb = torch.empty(J, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        rc = _lib.lib.mlgnn_linear_f32x3_bwd(go.data_ptr(), x.data_ptr(), weight.data_ptr(), _lib.ptr(gx), gw.data_ptr(),
                                             _lib.ptr(gb), ws.data_ptr(), ws.numel(), N, R, J,
                                             torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mlgnn_linear_f32x3_bwd")
        return (gx[:N] if gx is not None else None), gw, gb


WIDE_F32 = os.environ.get("MLGNN_WIDE_F32", "1") != "0"
SKINNY_MIN_K = 8192            # below this the weight is a few MB and the library's GEMM is not a stream problem


class _SkinnyLinear(torch.autograd.Function):
    """``y = x W^T + b`` for at most 64 rows with a very long input (the first Linear of MultilevelGNN's head,
    multilevel_gnn.py:121-127: ``[B, 84 096] -> 512`` at config/kirc.yaml): forward, input gradient and weight gradient as
    streams over the weight (``mlgnn_skinny_linear_fwd`` / ``_bwd``, csrc/skinny.hip).  The weight gradient is written
    straight into the parameter's slot of the flat gradient bucket when there is one (``mlgnn.dist.FlatGradBucket``): the
    172 MB tensor is then not copied again by ``collect()``."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x, w = _aligned(x), _aligned(weight)
        M, K = x.shape
        J = w.shape[0]
        y = torch.empty((M, J), dtype=torch.float32, device=x.device)
        n = int(_lib.lib.mlgnn_skinny_linear_fwd_workspace_floats(M, J, K))
        ws = torch.empty(n, dtype=torch.float32, device=x.device)
        b = bias.contiguous() if bias is not None else None
        rc = _lib.lib.mlgnn_skinny_linear_fwd(x.data_ptr(), w.data_ptr(), _lib.ptr(b), y.data_ptr(), ws.data_ptr(), n, M, J, K,
                                              torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mlgnn_skinny_linear_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.slot, ctx.param = getattr(weight, "_mlgnn_grad_slot", None), weight
        return y

    @staticmethod
    def backward(ctx, go):
        x, w = ctx.saved_tensors
        M, K = x.shape
        J = w.shape[0]
        go = go.contiguous()
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            slot = ctx.slot
            # (only when this backward DEFINES the gradient -- .grad released before the step; a gradient that is being
            # accumulated into must arrive in memory of its own)
            if (slot is not None and ctx.param.grad is None and slot.numel() == J * K and slot.dtype == torch.float32
                    and slot.device == x.device and slot.data_ptr() % 16 == 0):
                gw = slot.view(J, K)                      # (a fresh alias: autograd adopts it as .grad without a copy)
            else:
                gw = torch.empty((J, K), dtype=torch.float32, device=x.device)
        gb = torch.empty(J, dtype=torch.float32, device=x.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        rc = _lib.lib.mlgnn_skinny_linear_bwd(go.data_ptr(), x.data_ptr(), w.data_ptr(), _lib.ptr(gx), _lib.ptr(gw),
                                              _lib.ptr(gb), M, J, K, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mlgnn_skinny_linear_bwd")
        return gx, gw, gb


class _NarrowLinear(torch.autograd.Function):
    """``y = x W^T + b`` for tall fp32 rows with 1..8 input columns (the node encoder ``Linear(3, hidden)``,
    deepergcn.py:199-210): forward and weight / bias gradient as single streams over the ``[N, J]`` tensor
    (``mlgnn_narrow_linear_fwd`` / ``_bwd``, csrc/sage.hip).  The input gradient (the raw node features are data; asked
    for only by attribution tools) is a plain product."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x, w = x.contiguous(), weight.contiguous()
        N, R = x.shape
        J = w.shape[0]
        y = torch.empty((N, J), dtype=torch.float32, device=x.device)
        b = bias.contiguous() if bias is not None else None
        rc = _lib.lib.mlgnn_narrow_linear_fwd(x.data_ptr(), w.data_ptr(), _lib.ptr(b), y.data_ptr(), N, R, J,
                                              torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mlgnn_narrow_linear_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, go):
        x, w = ctx.saved_tensors
        N, R = x.shape
        J = w.shape[0]
        go = _aligned(go)
        gw = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            n = int(_lib.lib.mlgnn_narrow_linear_bwd_workspace_floats(R, J))
            ws = torch.empty(n, dtype=torch.float32, device=x.device)
            out = torch.empty(J * R + J, dtype=torch.float32, device=x.device)
            rc = _lib.lib.mlgnn_narrow_linear_bwd(go.data_ptr(), x.data_ptr(), out.data_ptr(), ws.data_ptr(), n, N, R, J,
                                                  torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "mlgnn_narrow_linear_bwd")
            gw, gb = out[:J * R].view(J, R), (out[J * R:] if ctx.has_bias else None)
        gx = go.matmul(w) if ctx.needs_input_grad[0] else None
        return gx, gw, gb


def linear(x, weight, bias=None, residual=None):
    """``nn.Linear`` forward (+ ``residual``: the identity branch of a residual block, added in the GEMM
    epilogue) with the tall-matrix kernels behind it when they apply (2-D fp32 CUDA input, >= 8192 rows,
    <= 32 output tiles of 32x32); ``F.linear`` otherwise."""
    if (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2
            and x.shape[1] >= SKINNY_MIN_K and torch.is_grad_enabled()
            and _lib.lib.mlgnn_skinny_linear_supported(x.shape[0], weight.shape[0], x.shape[1])):
        out = _SkinnyLinear.apply(x, weight, bias)
        return out if residual is None else out + residual
    if (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2
            and x.shape[0] >= WGRAD_MIN_ROWS and x.shape[1] <= 8
            and _lib.lib.mlgnn_narrow_linear_supported(x.shape[0], x.shape[1], weight.shape[0])):
        out = _NarrowLinear.apply(x, weight, bias)
        return out if residual is None else out + residual
    if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= WGRAD_MIN_ROWS
            and x.is_contiguous() and torch.is_grad_enabled()
            and _lib.lib.mlgnn_linear_wgrad_workspace_floats(x.shape[0], weight.shape[0], weight.shape[1], 0) > 0):
        return _TallLinear.apply(x, weight, bias, residual)
    if (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.dim() == 2
            and x.shape[0] >= WGRAD_MIN_ROWS and x.is_contiguous()
            and tall_matmul_supported(x.shape[0], x.shape[1], weight.shape[0], x.dtype)
            and tall_matmul_supported(x.shape[0], weight.shape[0], x.shape[1], x.dtype)):
        return _TallLinear.apply(x, weight, bias, residual)
    if (WIDE_F32 and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2
            and x.shape[0] >= WGRAD_MIN_ROWS and torch.is_grad_enabled()
            and _lib.lib.mlgnn_linear_f32x3_supported(x.shape[0], x.shape[1], weight.shape[0])):
        # fp32 widths past the tall kernels (hidden 512): three-term bf16 products instead of the library's fp32 GEMMs
        out = _WideLinearF32.apply(x, weight, bias)
        return out if residual is None else out + residual
    out = F.linear(x, weight, bias)
    return out if residual is None else out + residual


class _DenseSageFused(torch.autograd.Function):
    """One fused fp32-MFMA launch per direction (``mlgnn_dense_sage_fwd`` / ``_bwd``), one workgroup per pooled graph.
    fp32 or bf16 storage (all of x, adj, the weights in one type; arithmetic fp32, one rounding at each store)."""

    @staticmethod
    def forward(ctx, x, adj, w_rel, w_root, bias, normalize):
        B, n, C = x.shape
        O = w_rel.shape[0]
        x, adj, w_rel, w_root = x.contiguous(), adj.contiguous(), w_rel.contiguous(), w_root.contiguous()
        batched = adj.dim() == 3 and adj.shape[0] == B and B > 1
        y = torch.empty((B, n, O), dtype=x.dtype, device=x.device)
        rinv = torch.empty((B, n), dtype=torch.float32, device=x.device)
        ctx.bias_dtype = bias.dtype if bias is not None else None
        bias32 = f32_cached(bias) if bias is not None else None           # the kernels add the bias in fp32
        rc = _lib.lib.mlgnn_dense_sage_fwd(x.data_ptr(), adj.data_ptr(), w_rel.data_ptr(), w_root.data_ptr(),
                                           _lib.ptr(bias32), y.data_ptr(),
                                           rinv.data_ptr(), B, n, C, O, int(batched), int(normalize), _dt(x),
                                           torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mlgnn_dense_sage_fwd")
        ctx.save_for_backward(x, adj, w_rel, w_root, y, rinv)
        ctx.cfg = (batched, bool(normalize), bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, adj, w_rel, w_root, y, rinv = ctx.saved_tensors
        batched, normalize, has_bias = ctx.cfg
        B, n, C = x.shape
        O = w_rel.shape[0]
        gy = gy.contiguous()
        need_adj = ctx.needs_input_grad[1]
        gx = torch.empty_like(x)
        gadj = torch.empty((B, n, n), dtype=x.dtype, device=x.device) if need_adj else None
        gw = torch.empty(2 * O * C + O, dtype=torch.float32, device=x.device)
        ws_n = int(_lib.lib.mlgnn_dense_sage_bwd_workspace_floats(B, C, O))
        ws = torch.empty(ws_n, dtype=torch.float32, device=x.device)
        rc = _lib.lib.mlgnn_dense_sage_bwd(gy.data_ptr(), y.data_ptr(), rinv.data_ptr(), x.data_ptr(), adj.data_ptr(),
                                           w_rel.data_ptr(), w_root.data_ptr(), gx.data_ptr(), _lib.ptr(gadj),
                                           gw.data_ptr(), ws.data_ptr(), ws_n, B, n, C, O, int(batched),
                                           int(normalize), _dt(x), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mlgnn_dense_sage_bwd")
        if need_adj and not batched:
            gadj = gadj.float().sum(0, keepdim=True).to(adj.dtype).reshape(adj.shape)          # shared adjacency
        elif need_adj:
            gadj = gadj.reshape(adj.shape)
        if x.dtype != torch.float32:
            gw = gw.to(x.dtype)                                           # (weight gradients: fp32 sums, one converting copy)
        g_rel = gw[:O * C].view(O, C)
        g_root = gw[O * C:2 * O * C].view(O, C)
        g_b = gw[2 * O * C:].to(ctx.bias_dtype) if has_bias else None
        return gx, gadj, g_rel, g_root, g_b, None


def dense_sage(x, adj, w_rel, w_root, b_root, normalize=True):
    """``normalize(W_rel (A x / clamp(rowsum A, 1)) + W_root x + b)``; 2-D ``adj`` broadcasts.
    Pooled graphs of up to 160 nodes / 128 input / 64 output channels run as one fused launch."""
    x = x.unsqueeze(0) if x.dim() == 2 else x
    adj = adj.unsqueeze(0) if adj.dim() == 2 else adj
    B, n, c = x.shape
    if (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and adj.shape[0] in (1, B)
            and (adj.dtype != x.dtype or w_rel.dtype != x.dtype or w_root.dtype != x.dtype)):
        # (mixed storage types, e.g. an fp32 adjacency buffer in a bf16 model: one type for the kernel)
        adj, w_rel, w_root = adj.to(x.dtype), w_rel.to(x.dtype), w_root.to(x.dtype)
    if (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and adj.dtype == x.dtype and adj.shape[0] in (1, B)
            and adj.shape[-1] == n and adj.shape[-2] == n
            and _lib.lib.mlgnn_dense_sage_supported(n, c, w_rel.shape[0], int(adj.requires_grad))):
        return _DenseSageFused.apply(x, adj, w_rel, w_root, b_root, bool(normalize))
    agg = torch.matmul(adj, x) / adj.sum(dim=-1, keepdim=True).clamp(min=1)
    # both Linears see the [B*n, c] node rows: tall enough for the split-row weight-gradient kernel
    # (the library's TN GEMM for a 32..64-wide output over 56k rows runs on a handful of workgroups)
    out = linear(agg.reshape(-1, c), w_rel) + linear(x.reshape(-1, c), w_root, b_root)
    out = out.view(B, n, -1)
    return F.normalize(out, p=2.0, dim=-1) if normalize else out


class _DiffPoolFused(torch.autograd.Function):
    """Forward and backward are one fused fp32-MFMA launch each (``mlgnn_diffpool_fwd`` / ``_bwd``); fp32 or bf16 storage."""

    @staticmethod
    def forward(ctx, z, adj, s):
        B, N, C = z.shape
        K = s.shape[2]
        z, s, adj = z.contiguous(), s.contiguous(), adj.contiguous()
        batched = adj.dim() == 3 and adj.shape[0] == B and B > 1
        adj_k = adj if batched else adj.reshape(N, N)
        S = torch.empty_like(s)
        x_out = torch.empty((B, K, C), dtype=z.dtype, device=z.device)
        a_out = torch.empty((B, K, K), dtype=z.dtype, device=z.device)
        partial = torch.empty((B, 2), dtype=torch.float32, device=z.device)
        rc = _lib.lib.mlgnn_diffpool_fwd(z.data_ptr(), adj_k.data_ptr(), s.data_ptr(), S.data_ptr(),
                                         x_out.data_ptr(), a_out.data_ptr(), partial.data_ptr(), B, N, K, C,
                                         int(batched), _dt(z), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mlgnn_diffpool_fwd")
        tot = partial.sum(0)
        norm = torch.sqrt(tot[0])
        link = norm / adj.numel()
        ent = tot[1] / (B * N)
        ctx.save_for_backward(z, adj, S, norm)
        return x_out, a_out, link.to(z.dtype), ent.to(z.dtype)

    @staticmethod
    def backward(ctx, gx, ga, g_link, g_ent):
        z, adj, S, norm = ctx.saved_tensors
        B, N, C = z.shape
        K = S.shape[2]
        batched = adj.dim() == 3 and adj.shape[0] == B and B > 1
        coef = torch.stack([g_link.float() / (adj.numel() * norm), g_ent.float() / (B * N)]).contiguous()
        gz = torch.empty_like(z)
        gs = torch.empty_like(S)
        need_adj = ctx.needs_input_grad[1]
        gadj = torch.empty((B, N, N), dtype=z.dtype, device=z.device) if need_adj else None
        gx, ga = gx.to(z.dtype).contiguous(), ga.to(z.dtype).contiguous()
        rc = _lib.lib.mlgnn_diffpool_bwd(z.data_ptr(), adj.data_ptr(), S.data_ptr(), gx.data_ptr(), ga.data_ptr(),
                                         coef.data_ptr(), gz.data_ptr(), gs.data_ptr(), _lib.ptr(gadj), B, N, K, C,
                                         int(batched), _dt(z), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mlgnn_diffpool_bwd")
        if need_adj and not batched:
            gadj = gadj.float().sum(0, keepdim=True).to(adj.dtype).reshape(adj.shape)          # shared adjacency
        return gz, gadj, gs


class _DiffPoolLarge(torch.autograd.Function):
    """Pooled graphs past the fused small-graph kernel (BASELINE configs[4]: 4096 nodes, 1024 clusters): the chain of
    large bf16 products on the matrix cores, ``mlgnn_diffpool_large_fwd`` / ``_bwd`` (csrc/diffpool_large.hip).  A batch
    ``z [B,N,C]``, ``s [B,N,K]``, ``adj [B,N,N]`` or ``[1,N,N]`` (shared) runs as one grouped launch per step of the chain."""

    @staticmethod
    def forward(ctx, z, adj, s, adj_symmetric):
        B, N, C = z.shape
        K = s.shape[2]
        dev = z.device
        zb = z.contiguous() if z.dtype == torch.bfloat16 else z.to(torch.bfloat16).contiguous()
        ab = adj.contiguous() if adj.dtype == torch.bfloat16 else adj.to(torch.bfloat16).contiguous()
        s = s.contiguous()
        out_dtype = z.dtype
        adj_batched = int(adj.shape[0] == B and B > 1)
        S = torch.empty((B, N, K), dtype=torch.bfloat16, device=dev)
        x_out = torch.empty((B, K, C), dtype=out_dtype, device=dev)
        a_out = torch.empty((B, K, K), dtype=out_dtype, device=dev)
        stats = torch.empty(3, dtype=torch.float32, device=dev)
        scal = torch.empty(2, dtype=out_dtype, device=dev)
        ws = torch.empty(B * int(_lib.lib.mlgnn_diffpool_large_workspace_bytes(N, K, C)), dtype=torch.uint8, device=dev)
        rc = _lib.lib.mlgnn_diffpool_large_fwd(zb.data_ptr(), ab.data_ptr(), s.data_ptr(), _dt(s), S.data_ptr(),
                                               x_out.data_ptr(), a_out.data_ptr(), scal.data_ptr(), _dt(x_out),
                                               stats.data_ptr(), ws.data_ptr(), ws.numel(), N, K, C, B, adj_batched,
                                               torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mlgnn_diffpool_large_fwd")
        ctx.save_for_backward(zb, ab, s, S, ws, stats)
        ctx.cfg = (bool(adj_symmetric), z.dtype, adj_batched)
        ctx.adj_dtype, ctx.adj_shape = adj.dtype, adj.shape
        return x_out, a_out, scal[0], scal[1]

    @staticmethod
    def backward(ctx, gx, ga, g_link, g_ent):
        zb, ab, s, S, ws, stats = ctx.saved_tensors
        sym, z_dtype, adj_batched = ctx.cfg
        B, N, C = zb.shape
        K = S.shape[2]
        dev = zb.device
        gdt = torch.float32 if gx.dtype == torch.float32 else torch.bfloat16
        gx, ga = gx.to(gdt).contiguous(), ga.to(gdt).contiguous()
        if g_link.dtype != g_ent.dtype or g_link.dtype not in (torch.float32, torch.bfloat16):
            g_link, g_ent = g_link.float(), g_ent.float()
        gz = torch.empty((B, N, C), dtype=s.dtype, device=dev)
        gs = torch.empty((B, N, K), dtype=s.dtype, device=dev)
        # the adjacency of a second pooling level is the first level's S^T A S (diff_pooling.py:116-127): its gradient
        # is two more products of the same chain
        gadj = torch.empty((B, N, N), dtype=s.dtype, device=dev) if ctx.needs_input_grad[1] else None
        wb = torch.empty(B * int(_lib.lib.mlgnn_diffpool_large_bwd_workspace_bytes(N, K, C, int(sym))), dtype=torch.uint8,
                         device=dev)
        rc = _lib.lib.mlgnn_diffpool_large_bwd(zb.data_ptr(), ab.data_ptr(), s.data_ptr(), _dt(s), S.data_ptr(), ws.data_ptr(),
                                               gx.data_ptr(), ga.data_ptr(), _dt(gx), g_link.data_ptr(), g_ent.data_ptr(),
                                               _dt(g_link), stats.data_ptr(), gz.data_ptr(), gs.data_ptr(), _lib.ptr(gadj),
                                               int(sym), wb.data_ptr(), wb.numel(), N, K, C, B, adj_batched,
                                               torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mlgnn_diffpool_large_bwd")
        if gadj is not None:
            if not adj_batched and B > 1:
                gadj = gadj.float().sum(0, keepdim=True)         # shared adjacency: the sum over the graphs that read it
            gadj = gadj.to(ctx.adj_dtype).reshape(ctx.adj_shape)
        return gz.to(z_dtype), gadj, gs, None


def _dt(t):
    return 1 if t.dtype == torch.bfloat16 else 0


class _DiffPoolLargeFP32(torch.autograd.Function):
    """fp32 inputs at sizes past the fused small-graph kernel (multiples of 128): the product chain of
    :class:`_DiffPoolLarge` with every product as three bf16 terms on the matrix cores (fp32-level accuracy: within 1e-4
    of fp64), one C entry point each way (``mlgnn_diffpool_large_f32_fwd`` / ``_bwd``, csrc/diffpool_large.hip); a batch
    ``z [B,N,C]``, ``s [B,N,K]``, ``adj [B,N,N]`` or ``[1,N,N]`` (shared) runs as grouped launches."""

    @staticmethod
    def forward(ctx, z, adj, s, adj_symmetric):
        B, N, C = z.shape
        K = s.shape[2]
        dev = z.device
        z, adj, s = z.contiguous(), adj.contiguous(), s.contiguous()
        adj_batched = int(adj.shape[0] == B and B > 1)
        S = torch.empty((B, N, K), dtype=torch.float32, device=dev)
        x_out = torch.empty((B, K, C), dtype=torch.float32, device=dev)
        a_out = torch.empty((B, K, K), dtype=torch.float32, device=dev)
        stats = torch.empty(3, dtype=torch.float32, device=dev)
        scal = torch.empty(2, dtype=torch.float32, device=dev)
        ws = torch.empty(B * int(_lib.lib.mlgnn_diffpool_large_f32_workspace_bytes(N, K, C)), dtype=torch.uint8, device=dev)
        rc = _lib.lib.mlgnn_diffpool_large_f32_fwd(z.data_ptr(), adj.data_ptr(), s.data_ptr(), S.data_ptr(), x_out.data_ptr(),
                                                   a_out.data_ptr(), scal.data_ptr(), stats.data_ptr(), ws.data_ptr(),
                                                   ws.numel(), N, K, C, B, adj_batched,
                                                   torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mlgnn_diffpool_large_f32_fwd")
        ctx.save_for_backward(adj, s, ws, stats)
        ctx.cfg = (bool(adj_symmetric), adj_batched, (B, N, K, C))
        ctx.adj_shape = adj.shape
        return x_out, a_out, scal[0], scal[1]

    @staticmethod
    def backward(ctx, gx, ga, g_link, g_ent):
        adj, s, ws, stats = ctx.saved_tensors
        sym, adj_batched, (B, N, K, C) = ctx.cfg
        dev = s.device
        gx, ga = gx.float().contiguous(), ga.float().contiguous()
        g_link, g_ent = g_link.float().contiguous(), g_ent.float().contiguous()
        gz = torch.empty((B, N, C), dtype=torch.float32, device=dev)
        gs = torch.empty((B, N, K), dtype=torch.float32, device=dev)
        # the adjacency of a second pooling level is the first level's S^T A S (diff_pooling.py:116-127)
        gadj = torch.empty((B, N, N), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        wb = torch.empty(B * int(_lib.lib.mlgnn_diffpool_large_f32_bwd_workspace_bytes(N, K, C, int(sym))), dtype=torch.uint8,
                         device=dev)
        rc = _lib.lib.mlgnn_diffpool_large_f32_bwd(adj.data_ptr(), s.data_ptr(), ws.data_ptr(), gx.data_ptr(), ga.data_ptr(),
                                                   g_link.data_ptr(), g_ent.data_ptr(), stats.data_ptr(), gz.data_ptr(),
                                                   gs.data_ptr(), _lib.ptr(gadj), int(sym), wb.data_ptr(), wb.numel(),
                                                   N, K, C, B, adj_batched, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mlgnn_diffpool_large_f32_bwd")
        if gadj is not None:
            if not adj_batched and B > 1:
                gadj = gadj.sum(0, keepdim=True)                 # shared adjacency: the sum over the graphs that read it
            gadj = gadj.reshape(ctx.adj_shape)
        return gz, gadj, gs, None


def diff_pool_large_supported(z, adj, s):
    B, N, C = z.shape
    same = z.dtype == s.dtype == adj.dtype and z.dtype in (torch.bfloat16, torch.float32)
    return z.is_cuda and same and bool(_lib.lib.mlgnn_diffpool_large_supported(N, s.shape[2], C))


def _diff_pool_large(z, adj, s, adj_symmetric=False):
    """The whole batch through the grouped launches of the C entry points: bf16 storage on the bf16 chain, fp32 on its
    three-term form.  Losses as the reference combines them (one Frobenius norm over the batch, entropy over all nodes)."""
    fn = _DiffPoolLarge if z.dtype == torch.bfloat16 else _DiffPoolLargeFP32
    return fn.apply(z, adj, s, adj_symmetric)


def _diff_pool_library(z, adj, s):
    s = torch.softmax(s, dim=-1)
    st = s.transpose(1, 2)
    out = torch.matmul(st, z)
    out_adj = torch.matmul(torch.matmul(st, adj), s)
    link = torch.norm(adj - torch.matmul(s, st), p=2) / adj.numel()
    ent = (-s * torch.log(s + DIFFPOOL_EPS)).sum(dim=-1).mean()
    return out, out_adj, link, ent


def dense_diff_pool(z, adj, s, adj_symmetric=False):
    """``S = softmax(s)``; returns ``(S^T Z, S^T A S, ||A - S S^T||_F / numel(A), mean entropy)``.
    Pooled graphs of up to 160 nodes / 48 clusters / 64 channels (the reference's 146 -> 37 -> 10)
    run as one fused fp32-MFMA launch; graphs whose sizes are multiples of 128 (BASELINE configs[4]:
    4096 nodes, 1024 clusters) as the matrix-core product chain of csrc/diffpool_large.hip (bf16) or its three-term
    fp32-accurate form; anything else as batched library GEMMs.  ``adj_symmetric`` promises ``adj == adj^T`` (saves a third of the large backward)."""
    z = z.unsqueeze(0) if z.dim() == 2 else z
    adj = adj.unsqueeze(0) if adj.dim() == 2 else adj
    s = s.unsqueeze(0) if s.dim() == 2 else s
    B, N, C = z.shape
    K = s.shape[2]
    if (z.is_cuda and z.dtype in (torch.float32, torch.bfloat16) and adj.shape[0] in (1, B)
            and _lib.lib.mlgnn_diffpool_fwd_supported(N, K, C)):
        # fp32 or bf16 storage (one type for z / adj / logits; fp32 arithmetic, one rounding at each store)
        return _DiffPoolFused.apply(z, adj.to(z.dtype), s.to(z.dtype))
    if adj.shape[0] in (1, B) and diff_pool_large_supported(z, adj, s):
        return _diff_pool_large(z, adj, s, adj_symmetric)
    return _diff_pool_library(z, adj, s)
