"""Fused LayerNorm (+ ReLU) (reference: ``norm_layer('layer')`` + ``act_layer('relu')`` chained by
``MLP`` -- ``models/gcn_lib/sparse/torch_nn.py:27-38,54-75`` -- and by the res+ block,
``models/deepergcn.py:236-241``).  One HIP pass forward, one backward (``csrc/norm.hip``)."""
import torch
import torch.nn.functional as F

from . import _lib
from .ops import _DTYPE_IDS, DTYPE_F32, _stream, tag_row_max


def _dtype_id(t):
    return 1 if t.dtype == torch.bfloat16 else 0


def fused_supported(x):
    """fp32: d <= 256 with d % 4 == 0, or d <= 512 with d % 8 == 0; bf16 storage (fp32 arithmetic): d <= 512, d % 8 == 0."""
    if not (x.is_cuda and x.dim() == 2):
        return False
    if x.dtype == torch.float32:
        d = x.shape[1]
        return (0 < d <= 256 and d % 4 == 0) or (256 < d <= 512 and d % 8 == 0)
    return x.dtype == torch.bfloat16 and 0 < x.shape[1] <= 512 and x.shape[1] % 8 == 0


def _f32_params(weight, bias):
    """gamma / beta as the kernels take them: fp32, contiguous (a bf16 model keeps bf16 parameters: [d] casts)."""
    from .ops import f32_cached
    return f32_cached(weight), f32_cached(bias)


def _keep_mask(x, p):
    """Dropout keep flags (one byte per element) and the 1/(1-p) scale; ``(None, 1.0)`` for p == 0."""
    if not p:
        return None, 1.0
    if p >= 1.0:
        return torch.zeros(x.shape, dtype=torch.uint8, device=x.device), 0.0
    return torch.empty(x.shape, dtype=torch.uint8, device=x.device).bernoulli_(1.0 - p), 1.0 / (1.0 - p)


class _LayerNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, relu, keep=None, keep_scale=1.0):
        x = x.contiguous()
        rows, d = x.shape
        out = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        ctx.param_dtype = weight.dtype
        weight, bias = _f32_params(weight, bias)
        # max |row| rides along for the fp32 split-precision GEMM that consumes the result; bf16 has no use for it
        ctx.row_max = torch.empty(rows, dtype=torch.float32, device=x.device) if x.dtype == torch.float32 else None
        if keep is not None and (keep.shape != x.shape or keep.dtype != torch.uint8 or not keep.is_contiguous()):
            raise ValueError("dropout keep mask must be a contiguous uint8 tensor of the input's shape")
        rc = _lib.lib.mlgnn_layernorm_act_fwd(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                              mean.data_ptr(), rstd.data_ptr(), _lib.ptr(ctx.row_max), _lib.ptr(keep),
                                              float(keep_scale), rows, d,
                                              float(eps), int(relu), _DTYPE_IDS[x.dtype], _stream())
        _lib.check(rc, "mlgnn_layernorm_act_fwd")
        ctx.relu = bool(relu)
        ctx.keep, ctx.keep_scale = keep, float(keep_scale)
        ctx.save_for_backward(x, weight, bias, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, go):
        gx, ggb = _ln_backward(ctx, go, None)
        return gx, ggb[0], ggb[1], None, None, None, None


def _ln_backward(ctx, go, extra):
    x, weight, bias, mean, rstd = ctx.saved_tensors
    rows, d = x.shape
    go = go.contiguous()
    if extra is not None:
        extra = extra.contiguous()
    gx = torch.empty_like(x)
    ggb = torch.empty((2, d), dtype=torch.float32, device=x.device)
    dt = _DTYPE_IDS[x.dtype]
    n = int(_lib.lib.mlgnn_layernorm_bwd_workspace_floats(rows, d, dt))
    ws = torch.empty(n, dtype=torch.float32, device=x.device)
    row_max = torch.empty(rows, dtype=torch.float32, device=x.device) if x.dtype == torch.float32 else None
    rc = _lib.lib.mlgnn_layernorm_act_bwd(go.data_ptr(), x.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                          mean.data_ptr(), rstd.data_ptr(), _lib.ptr(extra), gx.data_ptr(),
                                          _lib.ptr(row_max), ggb.data_ptr(), ws.data_ptr(), n, _lib.ptr(ctx.keep),
                                          ctx.keep_scale, rows, d,
                                          int(ctx.relu), dt, _stream())
    _lib.check(rc, "mlgnn_layernorm_act_bwd")
    if row_max is not None:
        tag_row_max(gx, row_max)               # the gradient usually goes straight into a Linear's backward GEMM
    return gx, ggb.to(ctx.param_dtype)


def ln_backward_saved(go, x, weight, bias, mean, rstd, relu, extra=None):
    """LayerNorm(+ReLU) backward from the saved input and statistics, outside an autograd node of its own (the fused
    MLP's post-LayerNorm, :class:`mlgnn.dense._FusedMLP2`): ``-> (grad_x (+ extra), grad_gamma, grad_beta)``; the
    result carries its row maxima for the GEMMs that consume it."""
    rows, d = x.shape
    go = go.contiguous()
    if extra is not None:
        extra = extra.contiguous()
    weight, bias = _f32_params(weight, bias)
    gx = torch.empty_like(x)
    ggb = torch.empty((2, d), dtype=torch.float32, device=x.device)
    dt = _DTYPE_IDS[x.dtype]
    n = int(_lib.lib.mlgnn_layernorm_bwd_workspace_floats(rows, d, dt))
    ws = torch.empty(n, dtype=torch.float32, device=x.device)
    row_max = torch.empty(rows, dtype=torch.float32, device=x.device) if x.dtype == torch.float32 else None
    rc = _lib.lib.mlgnn_layernorm_act_bwd(go.data_ptr(), x.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                          mean.data_ptr(), rstd.data_ptr(), _lib.ptr(extra), gx.data_ptr(),
                                          _lib.ptr(row_max), ggb.data_ptr(), ws.data_ptr(), n, None, 1.0, rows, d,
                                          int(relu), dt, _stream())
    _lib.check(rc, "mlgnn_layernorm_act_bwd")
    if row_max is not None:
        tag_row_max(gx, row_max)
    return gx, ggb[0], ggb[1]


def ln_backward_normalised(go, xhat, weight, bias, rstd, relu=True):
    """LayerNorm(+ReLU) backward when the stored activation is already normalised (``xhat``, ``rstd`` from the
    first GEMM of :class:`mlgnn.dense._FusedMLP2`): ``-> (grad_x, grad_gamma, grad_beta, max |grad_x| per row)``."""
    rows, d = xhat.shape
    go = go.contiguous()
    gx = torch.empty_like(xhat)
    ggb = torch.empty((2, d), dtype=torch.float32, device=xhat.device)
    n = int(_lib.lib.mlgnn_layernorm_bwd_workspace_floats(rows, d, DTYPE_F32))
    ws = torch.empty(n, dtype=torch.float32, device=xhat.device)
    row_max = torch.empty(rows, dtype=torch.float32, device=xhat.device)
    weight, bias = weight.contiguous(), bias.contiguous()
    rc = _lib.lib.mlgnn_layernorm_act_bwd(go.data_ptr(), xhat.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                          None, rstd.data_ptr(), None, gx.data_ptr(), row_max.data_ptr(),
                                          ggb.data_ptr(), ws.data_ptr(), n, None, 1.0, rows, d, int(relu), DTYPE_F32,
                                          _stream())
    _lib.check(rc, "mlgnn_layernorm_act_bwd")
    return gx, ggb[0], ggb[1], row_max


class _LayerNormActFork(torch.autograd.Function):
    """``(relu?(LayerNorm(x)), x)``: the second output is the input itself, handed back so that the gradient
    arriving on the identity branch of a residual block (``h = f(norm(h)) + h``, deepergcn.py:236-241) meets
    the LayerNorm gradient inside ONE backward pass instead of a separate accumulation kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, relu, keep=None, keep_scale=1.0):
        out = _LayerNormAct.forward(ctx, x, weight, bias, eps, relu, keep, keep_scale)
        return out, x.view_as(x)

    @staticmethod
    def backward(ctx, go, g_identity):
        gx, ggb = _ln_backward(ctx, go, g_identity)
        return gx, ggb[0], ggb[1], None, None, None, None


def layer_norm_act(x, weight, bias, eps=1e-5, relu=False, dropout_p=0.0, dropout_mask=None):
    """``dropout?(relu?(LayerNorm(x)))`` over the last dimension of a 2-D tensor -- norm, activation and the dropout
    the res+ block puts behind them (deepergcn.py:239-240,246-247) in one pass.  ``dropout_p``: probability (the
    caller passes 0 outside training); ``dropout_mask``: explicit uint8 keep flags instead of a fresh draw (tests).
    Shapes the fused kernel does not cover (see :func:`fused_supported`) take ATen on the same device."""
    if weight is not None and bias is not None and fused_supported(x):
        keep, scale = (dropout_mask, 1.0 / (1.0 - dropout_p)) if dropout_mask is not None else _keep_mask(x, dropout_p)
        return _tag_from_node(_LayerNormAct.apply(x, weight, bias, eps, relu, keep, scale))
    y = F.layer_norm(x, (x.shape[-1],), weight, bias, eps)
    y = F.relu(y) if relu else y
    if dropout_mask is not None:
        return y * dropout_mask.to(y.dtype) / (1.0 - dropout_p)
    return F.dropout(y, dropout_p, True) if dropout_p else y


def _tag_from_node(y):
    rm = getattr(y.grad_fn, "row_max", None) if y.grad_fn is not None else None
    return tag_row_max(y, rm) if rm is not None else y


def layer_norm_act_fork(x, weight, bias, eps=1e-5, relu=False, dropout_p=0.0, dropout_mask=None):
    """``(layer_norm_act(x), x)`` for a residual block: use the second value as the identity branch
    (``h = f(y) + x``) so that its gradient is added inside the LayerNorm backward kernel."""
    if weight is not None and bias is not None and fused_supported(x) and x.is_contiguous():
        keep, scale = (dropout_mask, 1.0 / (1.0 - dropout_p)) if dropout_mask is not None else _keep_mask(x, dropout_p)
        y, identity = _LayerNormActFork.apply(x, weight, bias, eps, relu, keep, scale)
        return _tag_from_node(y), identity
    return layer_norm_act(x, weight, bias, eps, relu, dropout_p, dropout_mask), x


class _MsgNormAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, m, scale):
        x, m = x.contiguous(), m.contiguous()
        rows, d = x.shape
        h = torch.empty_like(x)
        scale32 = scale if scale.dtype == torch.float32 else scale.float()       # the learnable scalar stays fp32
        rc = _lib.lib.mlgnn_msgnorm_add_fwd(x.data_ptr(), m.data_ptr(), scale32.data_ptr(), h.data_ptr(), rows, d,
                                            _dtype_id(x), _stream())
        _lib.check(rc, "mlgnn_msgnorm_add_fwd")
        ctx.save_for_backward(x, m, scale32)
        ctx.scale_dtype = scale.dtype
        return h

    @staticmethod
    def backward(ctx, gh):
        x, m, scale = ctx.saved_tensors
        rows, d = x.shape
        gh = gh.contiguous()
        gx, gm = torch.empty_like(x), torch.empty_like(m)
        gs = torch.empty(1, dtype=torch.float32, device=x.device)
        n = int(_lib.lib.mlgnn_msgnorm_bwd_workspace_floats(rows, d))
        ws = torch.empty(max(n, 1), dtype=torch.float32, device=x.device)
        rc = _lib.lib.mlgnn_msgnorm_add_bwd(gh.data_ptr(), x.data_ptr(), m.data_ptr(), scale.data_ptr(),
                                            gx.data_ptr(), gm.data_ptr(), gs.data_ptr(), ws.data_ptr(), n, rows, d,
                                            _dtype_id(x), _stream())
        _lib.check(rc, "mlgnn_msgnorm_add_bwd")
        return gx, gm, (gs.to(ctx.scale_dtype) if ctx.needs_input_grad[2] else None)


def msg_norm_add(x, m, scale):
    """``x + normalize(m, dim=1) * ||x|| * scale`` (MsgNorm + GENConv root add, torch_message.py:175-179,
    torch_vertex.py:86-89) in one HIP pass each way; ATen ops for widths the kernel does not cover."""
    if (m.shape == x.shape and m.dtype == x.dtype and x.is_cuda and x.dim() == 2
            and ((x.dtype == torch.float32 and 0 < x.shape[1] <= 256 and x.shape[1] % 4 == 0)
                 or (x.dtype == torch.bfloat16 and 0 < x.shape[1] <= 512 and x.shape[1] % 8 == 0))):
        return _MsgNormAdd.apply(x, m, scale)
    return x + F.normalize(m, p=2.0, dim=1) * x.norm(p=2, dim=1, keepdim=True) * scale
