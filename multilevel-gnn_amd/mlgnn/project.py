"""Gene -> pathway learnable-projection pooling (reference ``models/multilevel_gnn.py:212-239``).

``out[b, c, s, k] = sum_{g : raw_indice[b,g] = s} x[b*NN + match[b,g], c] * [match >= 0] * W[g,k]``

The reference materialises ``[B, G, C, k]`` three times (gather, repeat, permute) and reduces it
with an atomic ``scatter_reduce``.  Here the membership table is grouped by segment and by node
once (:class:`Membership`), and three atomic-free gather-reduce kernels do forward, input gradient
and weight gradient (``csrc/project.hip``).
"""
import torch

from . import _lib
from .ops import DTYPE_BF16, DTYPE_F32, _dev_f32, _stream


class Membership:
    """Index tables of one ``(gene_pca_match, raw_indice)`` pair (int32, device resident)."""

    def __init__(self, gene_pca_match, raw_indice, nodes_per_graph, n_segments, n_rows, match_mask=True):
        B, G = gene_pca_match.shape
        dev = gene_pca_match.device
        self.B, self.G, self.S, self.R = B, G, int(n_segments), int(n_rows)
        offs = torch.arange(B, device=dev)[:, None]
        row = gene_pca_match.long() + offs * nodes_per_graph
        if match_mask:
            row = torch.where(gene_pca_match >= 0, row, torch.full_like(row, -1))
        else:
            row = torch.remainder(row, n_rows)              # negative index wraps like the reference's x[idx]
        # an index past the last node row would be an IndexError in the reference; here it must never
        # become an out-of-bounds read, so it is treated as an absent member
        row = torch.where(row >= n_rows, torch.full_like(row, -1), row)
        seg = (raw_indice.long().clamp(0, n_segments - 1) + offs * n_segments).reshape(-1)
        row = row.reshape(-1)
        order = torch.sort(seg, stable=True).indices
        self.seg_mem = order.to(torch.int32)
        self.seg_ptr = self._ptr(seg, B * self.S)
        self.mem_seg = seg.to(torch.int32)
        self.mem_row = row.to(torch.int32)
        present = torch.nonzero(row >= 0).reshape(-1)
        order_n = torch.sort(row[present], stable=True).indices
        self.node_mem = present[order_n].to(torch.int32)
        self.node_ptr = self._ptr(row[present], self.R)

    @staticmethod
    def _ptr(index, n):
        ptr = torch.zeros(n + 1, dtype=torch.int64, device=index.device)
        torch.cumsum(torch.bincount(index, minlength=n), 0, out=ptr[1:])
        return ptr.to(torch.int32)


_MEMBERSHIP_CACHE = []


def membership_tables(gene_pca_match, raw_indice, nodes_per_graph, n_segments, n_rows, match_mask=True):
    """The table is a per-fold constant in the reference's data; batches hand in the same tensors
    again and again, so the last few results are kept (tensor identity + version)."""
    from .graph import _same_view
    key = (nodes_per_graph, n_segments, n_rows, bool(match_mask), gene_pca_match._version, raw_indice._version)
    for ent in _MEMBERSHIP_CACHE:
        # (any view of the same elements: a loader that expands ONE per-fold table to every batch hits)
        if _same_view(ent[0], gene_pca_match) and _same_view(ent[1], raw_indice) and ent[2] == key:
            return ent[3]
    m = Membership(gene_pca_match, raw_indice, nodes_per_graph, n_segments, n_rows, match_mask)
    _MEMBERSHIP_CACHE.insert(0, (gene_pca_match, raw_indice, key, m))
    del _MEMBERSHIP_CACHE[4:]
    return m


class _SegmentProject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, tables, n_groups=0):
        # rows in the model's storage type (fp32 or bf16: the kernels accumulate in fp32 either way and round once at
        # the store); the projection weights and their gradient partials stay fp32
        if x.dtype == torch.bfloat16:
            if not x.is_cuda:
                raise RuntimeError("x must be a CUDA tensor (no CPU fallback)")
            x = x.contiguous()
        else:
            x = _dev_f32(x, "x")
        ctx.w_dtype = w.dtype
        w = _dev_f32(w.float() if w.dtype != torch.float32 else w, "weights")
        dt = DTYPE_BF16 if x.dtype == torch.bfloat16 else DTYPE_F32
        R, C = x.shape
        G, K = w.shape
        if G != tables.G or R != tables.R:
            raise ValueError("membership table does not match the inputs")
        out_t = torch.empty((tables.B * tables.S, K, C), dtype=x.dtype, device=x.device)
        rc = _lib.lib.mlgnn_segment_project_fwd(
            x.data_ptr(), w.data_ptr(), tables.seg_ptr.data_ptr(), _lib.ptr(tables.seg_mem),
            tables.mem_row.data_ptr(), out_t.data_ptr(), tables.B * tables.S, C, G, K, tables.S, int(n_groups), dt,
            _stream())
        _lib.check(rc, "mlgnn_segment_project_fwd")
        ctx.tables = tables
        ctx.n_groups = int(n_groups)
        ctx.dt = dt
        ctx.save_for_backward(x, w)
        return out_t

    @staticmethod
    def backward(ctx, gout_t):
        x, w = ctx.saved_tensors
        t = ctx.tables
        R, C = x.shape
        G, K = w.shape
        gout_t = gout_t.to(x.dtype).contiguous() if ctx.dt == DTYPE_BF16 else _dev_f32(gout_t, "grad_out")
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gwp = torch.empty((t.B * G, K), dtype=torch.float32, device=x.device) if ctx.needs_input_grad[1] else None
        rc = _lib.lib.mlgnn_segment_project_bwd(
            gout_t.data_ptr(), x.data_ptr(), w.data_ptr(), t.seg_ptr.data_ptr(), _lib.ptr(t.seg_mem),
            t.mem_row.data_ptr(), t.mem_seg.data_ptr(), t.node_ptr.data_ptr(), _lib.ptr(t.node_mem),
            _lib.ptr(gx), _lib.ptr(gwp), t.B * t.S, R, C, G, K, t.S, ctx.n_groups, ctx.dt, _stream())
        _lib.check(rc, "mlgnn_segment_project_bwd")
        gw = gwp.reshape(t.B, G, K).sum(0).to(ctx.w_dtype) if gwp is not None else None
        return gx, gw, None, None


def segment_project(x_nodes, gene_pca_match, raw_indice, weights, nodes_per_graph, n_segments,
                    match_mask=True, pooled_groups=0):
    """``x_nodes [B*NN, C]`` -> ``[B, C, n_segments, k]``; ``weights [G, k]`` already carries the
    info mask.  ``match_mask=False`` wraps a negative ``match`` exactly as the reference's advanced
    indexing does.  ``pooled_groups = NG > 0``: the same numbers as the contiguous batch of pathway graphs
    ``[B * NG * k, n_segments / NG, C]`` -- what ``out.reshape(B, C, n_segments / NG, NG * k).permute(0, 3, 2, 1)
    .reshape(-1, n_segments / NG, C)`` (vae.py:238-243) yields, written by the kernel in that order (no transposing
    copy of the result or of its gradient)."""
    B = gene_pca_match.shape[0]
    tables = membership_tables(gene_pca_match, raw_indice, nodes_per_graph, n_segments, x_nodes.shape[0],
                               match_mask)
    C_in = x_nodes.shape[1]
    if x_nodes.dtype == torch.bfloat16 and C_in % 8 != 0 and C_in > 64:
        # (a bf16 width the 16-byte path does not take and the scalar weight-gradient kernel does not cover: fp32 rows)
        out_t = _SegmentProject.apply(x_nodes.float(), weights.float(), tables, int(pooled_groups)).to(torch.bfloat16)
    else:
        out_t = _SegmentProject.apply(x_nodes, weights, tables, int(pooled_groups))   # [B*S, k, C], storage type of x_nodes
    k, C = out_t.shape[1], out_t.shape[2]
    if pooled_groups:
        return out_t.reshape(B * pooled_groups * k, n_segments // pooled_groups, C)
    return out_t.reshape(B, n_segments, k, C).permute(0, 3, 1, 2)    # [B, C, S, k]
