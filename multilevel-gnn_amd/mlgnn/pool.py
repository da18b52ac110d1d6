"""Per-graph readout over the ``batch`` vector (reference: PyG ``global_{add,mean,max}_pool``
called from ``deepergcn.py:148-155,319``; torch_scatter semantics: an empty graph yields 0, max keeps
the first maximal row).  Forward: two deterministic HIP stages (``csrc/pool.hip``); backward: a
``[B,d] -> [N,d]`` row broadcast (sum / mean) or a scatter of ``[B,d]`` values (max)."""
import torch

from . import _lib
from .ops import DTYPE_F32, _stream

_KINDS = {"sum": 0, "add": 0, "mean": 1, "max": 2}


class _SegmentPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ptr, batch, kind):
        x = x.contiguous()
        N, d = x.shape
        B = ptr.numel() - 1
        out = torch.empty((B, d), dtype=torch.float32, device=x.device)
        argmax = torch.empty((B, d), dtype=torch.int32, device=x.device) if kind == 2 else None
        nbytes = int(_lib.lib.mlgnn_segment_pool_workspace_bytes(B, d))
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=x.device)
        rc = _lib.lib.mlgnn_segment_pool_fwd(x.data_ptr(), ptr.data_ptr(), out.data_ptr(), _lib.ptr(argmax),
                                             ws.data_ptr(), nbytes, B, d, kind, DTYPE_F32, _stream())
        _lib.check(rc, "mlgnn_segment_pool_fwd")
        ctx.kind, ctx.shape = kind, (N, d)
        ctx.save_for_backward(ptr, batch, argmax)
        return out

    @staticmethod
    def backward(ctx, go):
        ptr, batch, argmax = ctx.saved_tensors
        N, d = ctx.shape
        if ctx.kind == 2:
            gx = go.new_zeros((N, d))
            live = argmax >= 0
            cols = torch.arange(d, device=go.device).expand_as(argmax)
            gx[argmax[live].long(), cols[live]] = go[live]
            return gx, None, None, None
        if ctx.kind == 1:
            cnt = (ptr[1:] - ptr[:-1]).clamp(min=1).to(go.dtype)
            go = go / cnt[:, None]
        return go.index_select(0, batch), None, None, None


def global_pool(x, batch, kind, num_graphs=None):
    """``x [N, d]``, ``batch [N]`` sorted graph id per node -> ``[B, d]``.  ``num_graphs`` avoids the
    device->host sync of ``batch.max() + 1`` the reference pays."""
    B = int(num_graphs) if num_graphs is not None else int(batch.max().item()) + 1
    batch = batch.to(torch.long)
    if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.dim() == 2:
        ptr = torch.zeros(B + 1, dtype=torch.int64, device=x.device)
        torch.cumsum(torch.bincount(batch, minlength=B), 0, out=ptr[1:])
        # bf16 storage: the readout sums thousands of rows per graph, which a bf16 accumulator (ATen's index_add_
        # on bf16) cannot hold -- reduce in fp32 (one up-cast pass per step), round the [B, d] result once.
        # A width that is not a multiple of 4 (the kernel's 16-byte rows) is zero padded for the call: still the
        # deterministic two-stage kernel, never an atomic scatter.
        d = x.shape[1]
        xf = x.float()
        if d % 4:
            xf = torch.nn.functional.pad(xf, (0, 4 - d % 4))
        out = _SegmentPool.apply(xf, ptr.to(torch.int32), batch, _KINDS[kind])
        return out[:, :d].to(x.dtype)
    raise TypeError("global_pool wants a 2-D fp32 / bf16 CUDA tensor (the accelerated path has no CPU fallback)")
