"""One GraphSAGE layer of the shipped configs (``gnn_name: sage``, config/kirc.yaml / gbm.yaml) as a single autograd
node over the hand-written kernels.

Reference (models/gcn_lib/sparse/torch_vertex.py:269-304, torch_nn.py:54-75): per edge ``(x_j w_ij [- x_i]) W_r^T``,
mean over the incoming edges incl. the added self loop, ``nn(cat(x, aggr))`` with ``nn = Linear(in + out, out) ->
LeakyReLU(0.2)``; MultilevelGNN then multiplies the last layer's rows by the input value (multilevel_gnn.py:205-207).

Here, by linearity, with ``W_nn = [W_x | W_a]`` split at column ``in``::

    agg = weighted mean_j(x_j)                       csr_aggregate_fwd (mean and W_r commute: N rows instead of E + N)
    W_c = W_a W_r   (relative: W_x <- W_x - W_c)     [out, in] -- lin_r folded into the update's weight
    y   = leaky_relu([x | agg] [W_x | W_c]^T + b) * mask          ONE product, mlgnn_tallgemm_dual: neither
                                                                   aggr_out [N,out] nor cat(x, aggr_out) exists

backward: ``dz`` = LeakyReLU / mask backward (mlgnn_leaky_relu_bwd, with max |row|), ``d agg = dz W_c`` (tall GEMM),
the aggregation's transpose, ``dx = dz W_x + (that)`` (tall GEMM, the sum in its epilogue), ``dW = dz^T [x | agg]``
(two split-row weight-gradient products), and the chain rule of the fold on [out, in]-sized matrices."""
import torch

from . import _lib
from .dense import WGRAD_MIN_ROWS, _aligned, _wgrad, tall_matmul_nt, tall_matmul_supported
from .ops import AGGR_MEAN, AGGR_SUM, EDGE_NONE, MSG_IDENTITY, MSG_WEIGHTED, _DTYPE_IDS, _stream, tag_row_max

STATS = {"fused": 0, "fallback": 0}


def sage_layer_supported(x, w_nn, w_r, has_conv_bias):
    """fp32 CUDA rows tall enough for the tall kernels, ``in`` a multiple of 16 with ``2 in`` in {64, 128, 256},
    ``out`` in {32, 64, 128}, the reference's bias-free ``lin_r`` / SAGEConv (RSAGEConv passes ``bias=False``)."""
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and not has_conv_bias
            and x.shape[0] >= WGRAD_MIN_ROWS and torch.is_grad_enabled()):
        return False
    n, cin = x.shape
    cout = w_nn.shape[0]
    if w_nn.shape[1] != cin + cout or tuple(w_r.shape) != (cout, cin) or w_nn.dtype != torch.float32:
        return False
    return (bool(_lib.lib.mlgnn_tallgemm_dual_supported(n, cin, cin, cout))
            and tall_matmul_supported(n, cout, cin) and _lib.lib.mlgnn_linear_wgrad_workspace_floats(n, cout, cin, 0) > 0)


class _SageLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_nn, b_nn, w_r, row_scale, graph, ew_pair, slope, relative, folded_mean):
        x = _aligned(x)
        N, cin = x.shape
        cout = w_nn.shape[0]
        dev = x.device
        # neighbourhood mean (weighted; count = number of edges incl. the self loop).  folded_mean: the weights already
        # carry 1 / in-degree (CSRGraph.mean_edge_scalar) -- a weighted SUM
        agg = torch.empty_like(x)
        msg = MSG_WEIGHTED if ew_pair is not None else MSG_IDENTITY
        aggr = AGGR_SUM if folded_mean else AGGR_MEAN
        ew = ew_pair[0] if ew_pair is not None else None
        hub, hub_keep = graph.hub_arg("dst", cin)
        rc = _lib.lib.mlgnn_csr_aggregate_fwd(
            x.data_ptr(), graph.rowptr.data_ptr(), graph.col.data_ptr(), _lib.ptr(ew), None, None, None, None,
            agg.data_ptr(), None, None, None, None, N, cin, _DTYPE_IDS[x.dtype], msg, EDGE_NONE, 0, aggr, 1.0, 1.0,
            None, None, 0.0, 0, hub, _stream())
        _lib.check(rc, "mlgnn_csr_aggregate_fwd")
        # lin_r folded into the update's weight: [W_x (- W_c) | W_c], W_c = W_a W_r  (one small launch)
        w_nn_c, w_r_c = w_nn.contiguous(), w_r.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        w_cat, w_x1, w_c = torch.empty((cout, 2 * cin), **f32), torch.empty((cout, cin), **f32), torch.empty((cout, cin), **f32)
        rc = _lib.lib.mlgnn_sage_fold_fwd(w_nn_c.data_ptr(), w_r_c.data_ptr(), w_cat.data_ptr(), w_x1.data_ptr(), w_c.data_ptr(),
                                          cin, cout, int(bool(relative)), _stream())
        _lib.check(rc, "mlgnn_sage_fold_fwd")
        y = torch.empty((N, cout), dtype=torch.float32, device=dev)
        y_max = torch.empty(N, dtype=torch.float32, device=dev)
        a_max = torch.empty(N, dtype=torch.float32, device=dev)
        nbytes = int(_lib.lib.mlgnn_tallgemm_workspace_bytes(2 * cin, cout, 0))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        bias = b_nn.contiguous() if b_nn is not None else None
        rs = row_scale.reshape(-1).contiguous() if row_scale is not None else None
        rc = _lib.lib.mlgnn_tallgemm_dual(x.data_ptr(), agg.data_ptr(), w_cat.data_ptr(), _lib.ptr(bias), float(slope),
                                          _lib.ptr(rs), y.data_ptr(), y_max.data_ptr(), a_max.data_ptr(), ws.data_ptr(),
                                          nbytes, N, cin, cin, cout, _stream())
        _lib.check(rc, "mlgnn_tallgemm_dual")
        ctx.save_for_backward(x, agg, y, w_nn_c, w_r_c, w_x1, w_c, rs, a_max)
        ctx.graph, ctx.ew_pair = graph, ew_pair
        ctx.cfg = (float(slope), bool(relative), b_nn is not None, msg, aggr)
        ctx.mark_non_differentiable(y_max)
        return y, y_max

    @staticmethod
    def backward(ctx, gy, _g_max):
        x, agg, y, w_nn, w_r, w_x1, w_c, rs, a_max = ctx.saved_tensors
        slope, relative, has_bias, msg, aggr = ctx.cfg
        g = ctx.graph
        N, cin = x.shape
        cout = y.shape[1]
        gy = _aligned(gy)
        dz = torch.empty_like(y)
        dz_max = torch.empty(N, dtype=torch.float32, device=x.device)
        rc = _lib.lib.mlgnn_leaky_relu_bwd(gy.data_ptr(), y.data_ptr(), _lib.ptr(rs), slope, dz.data_ptr(), dz_max.data_ptr(),
                                           N, cout, _stream())
        _lib.check(rc, "mlgnn_leaky_relu_bwd")
        gx = None
        if ctx.needs_input_grad[0]:
            # d agg = dz W_c, through the transposed aggregation, then dx = dz W_x' + that (the sum in the GEMM's epilogue)
            dagg = tall_matmul_nt(dz, w_c, row_max=dz_max, bt_transposed=True)
            gagg = torch.empty_like(dagg)
            ew_t = ctx.ew_pair[1] if ctx.ew_pair is not None else None
            hub, hub_keep = g.hub_arg("src", cin)
            rc = _lib.lib.mlgnn_csr_aggregate_bwd(
                dagg.data_ptr(), None, None, None, None, g.rowptr_t.data_ptr(), g.col_t.data_ptr(), g.pos_t.data_ptr(),
                g.rowptr.data_ptr(), _lib.ptr(ew_t), None, None, None, None, None, gagg.data_ptr(), None, None, None, 0,
                N, cin, _DTYPE_IDS[dagg.dtype], msg, EDGE_NONE, 0, aggr, 0, 1.0, 1.0, None, None, 0.0, 0, 0, hub,
                None, None, _stream())
            _lib.check(rc, "mlgnn_csr_aggregate_bwd")
            gx = tall_matmul_nt(dz, w_x1, residual=gagg, row_max=dz_max, bt_transposed=True)
        # dW of the folded weight: dz^T x and dz^T agg (split-row kernels; a_max bounds both operands' rows)
        gw_x1, gb = _wgrad(dz, x, go_max=dz_max, x_max=a_max)
        gw_c, _ = _wgrad(dz, agg, go_max=dz_max, x_max=a_max)
        # chain rule of the fold (W_x' = W_x - rel W_c, W_c = W_a W_r): one small launch
        g_nn = torch.empty_like(w_nn)
        g_r = torch.empty_like(w_r)
        gw_x1, gw_c = gw_x1.contiguous(), gw_c.contiguous()
        rc = _lib.lib.mlgnn_sage_fold_bwd(gw_x1.data_ptr(), gw_c.data_ptr(), w_nn.data_ptr(), w_r.data_ptr(), g_nn.data_ptr(),
                                          g_r.data_ptr(), cin, cout, int(relative), _stream())
        _lib.check(rc, "mlgnn_sage_fold_bwd")
        return gx, g_nn, (gb if has_bias else None), g_r, None, None, None, None, None, None


def sage_layer(x, graph, edge_weight, w_nn, b_nn, w_r, slope=0.2, relative=False, row_scale=None):
    """``leaky_relu(cat(x, mean_j(x_j w_ij [- x_i]) W_r^T) W_nn^T + b_nn, slope) [* row_scale]`` -- see the module
    docstring; the caller checks :func:`sage_layer_supported`.  ``edge_weight``: [E] in COO order of ``graph`` or None.
    The result carries its row maxima for the next layer's kernels."""
    folded = bool(getattr(graph, "persistent", False))
    if folded:
        # a graph that outlives the step (the fold's topology): 1 / in-degree folded into the per-edge weights, once
        ew_pair = graph.mean_edge_scalar(edge_weight)
    else:
        # (by-destination / by-source copies of the weights: one launch, mlgnn_edge_table_to_csr)
        ew_pair = graph.edge_table(edge_weight, 1) if edge_weight is not None else None
    y, y_max = _SageLayer.apply(x, w_nn, b_nn, w_r, row_scale, graph, ew_pair, float(slope), bool(relative), folded)
    tag_row_max(y, y_max)
    STATS["fused"] += 1
    return y


class _NodeEmbed(torch.autograd.Function):
    """``h[b, n, :] = x[b, n] * E[n, :]`` (multilevel_gnn.py:151) with the row maxima of ``h``; gradient to ``E`` only
    (the input values are data)."""

    @staticmethod
    def forward(ctx, x, emb):
        nodes, C = emb.shape
        x = x.reshape(-1).contiguous()
        batch = x.numel() // nodes
        emb = _aligned(emb)
        h = torch.empty((batch * nodes, C), dtype=torch.float32, device=x.device)
        h_max = torch.empty(batch * nodes, dtype=torch.float32, device=x.device)
        rc = _lib.lib.mlgnn_node_embed_fwd(x.data_ptr(), emb.data_ptr(), h.data_ptr(), h_max.data_ptr(), batch, nodes, C,
                                           _stream())
        _lib.check(rc, "mlgnn_node_embed_fwd")
        ctx.save_for_backward(x)
        ctx.cfg = (batch, nodes, C)
        ctx.mark_non_differentiable(h_max)
        return h, h_max

    @staticmethod
    def backward(ctx, gh, _g):
        (x,) = ctx.saved_tensors
        batch, nodes, C = ctx.cfg
        gh = _aligned(gh)
        ge = torch.empty((nodes, C), dtype=torch.float32, device=gh.device)
        rc = _lib.lib.mlgnn_node_embed_bwd(x.data_ptr(), gh.data_ptr(), ge.data_ptr(), batch, nodes, C, _stream())
        _lib.check(rc, "mlgnn_node_embed_bwd")
        return None, ge


def node_embed_supported(x, emb):
    C = emb.shape[-1]
    lanes = C // 4
    return (x.is_cuda and x.dtype == torch.float32 and emb.dtype == torch.float32 and emb.dim() == 2 and not x.requires_grad
            and C % 4 == 0 and 0 < lanes <= 64 and lanes & (lanes - 1) == 0 and x.numel() % emb.shape[0] == 0
            and x.numel() < 2 ** 31)


def node_embed(x, emb):
    """-> ``[x.numel(), C]`` rows ``x[b, n] * emb[n]``, tagged with their row maxima."""
    h, h_max = _NodeEmbed.apply(x, emb)
    return tag_row_max(h, h_max)


class _LinearAct(torch.autograd.Function):
    """``leaky_relu(x W^T + b, slope)`` (slope 0: ReLU) for tall fp32 rows: the activation in the GEMM's epilogue
    (mlgnn_tallgemm_dual with one operand), its backward as ONE stream that also yields the row maxima the input- and
    weight-gradient products scale their operands with.  The 1x1 convolutions + ReLU of MultilevelGNN's pathway head
    (multilevel_gnn.py:98-104) over the [B * 146 * 3k] positions."""

    @staticmethod
    def forward(ctx, x, w, b, slope):
        x, w = _aligned(x), _aligned(w)
        N, R = x.shape
        J = w.shape[0]
        y = torch.empty((N, J), dtype=torch.float32, device=x.device)
        y_max = torch.empty(N, dtype=torch.float32, device=x.device)
        a_max = torch.empty(N, dtype=torch.float32, device=x.device)
        nbytes = int(_lib.lib.mlgnn_tallgemm_workspace_bytes(R, J, 0))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        bias = b.contiguous() if b is not None else None
        rc = _lib.lib.mlgnn_tallgemm_dual(x.data_ptr(), None, w.data_ptr(), _lib.ptr(bias), float(slope), None, y.data_ptr(),
                                          y_max.data_ptr(), a_max.data_ptr(), ws.data_ptr(), nbytes, N, R, 0, J, _stream())
        _lib.check(rc, "mlgnn_tallgemm_dual")
        ctx.save_for_backward(x, w, y, a_max)
        ctx.cfg = (float(slope), b is not None)
        ctx.mark_non_differentiable(y_max)
        return y, y_max

    @staticmethod
    def backward(ctx, gy, _g):
        x, w, y, a_max = ctx.saved_tensors
        slope, has_bias = ctx.cfg
        N, J = y.shape
        gy = _aligned(gy)
        dz = torch.empty_like(y)
        dz_max = torch.empty(N, dtype=torch.float32, device=y.device)
        rc = _lib.lib.mlgnn_leaky_relu_bwd(gy.data_ptr(), y.data_ptr(), None, slope, dz.data_ptr(), dz_max.data_ptr(), N, J,
                                           _stream())
        _lib.check(rc, "mlgnn_leaky_relu_bwd")
        gx = tall_matmul_nt(dz, w, row_max=dz_max, bt_transposed=True) if ctx.needs_input_grad[0] else None
        gw, gb = _wgrad(dz, x, go_max=dz_max, x_max=a_max)
        return gx, gw, (gb if has_bias else None), None


def linear_act_supported(x, w):
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
            and x.shape[0] >= WGRAD_MIN_ROWS and torch.is_grad_enabled() and w.dtype == torch.float32 and w.dim() == 2):
        return False
    n, r = x.shape
    j = w.shape[0]
    return (bool(_lib.lib.mlgnn_tallgemm_dual_supported(n, r, 0, j)) and tall_matmul_supported(n, j, r)
            and _lib.lib.mlgnn_linear_wgrad_workspace_floats(n, j, r, 0) > 0)


def linear_act(x, w, b, slope):
    """-> ``leaky_relu(x @ w.T + b, slope)`` tagged with its row maxima; the caller checks :func:`linear_act_supported`."""
    y, y_max = _LinearAct.apply(x, w, b, float(slope))
    return tag_row_max(y, y_max)


class _TransposeBatched(torch.autograd.Function):
    """``[B, R, C] -> [B, C, R]`` (fp32, contiguous both sides), its backward the same kernel the other way round."""

    @staticmethod
    def forward(ctx, x):
        B, R, C = x.shape
        y = torch.empty((B, C, R), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib.mlgnn_transpose_batched(x.data_ptr(), y.data_ptr(), B, R, C, _stream()), "mlgnn_transpose_batched")
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        B, C, R = gy.shape
        gx = torch.empty((B, R, C), dtype=torch.float32, device=gy.device)
        _lib.check(_lib.lib.mlgnn_transpose_batched(gy.data_ptr(), gx.data_ptr(), B, C, R, _stream()), "mlgnn_transpose_batched")
        return gx


def flatten_channel_last(x):
    """``torch.flatten(x, start_dim=1)`` of a ``[B, C, H, W]`` tensor that lives channel-last in memory (the layout the head's
    1x1 convolutions compute in, models.multilevel_gnn.HeadConv2d): one tiled transpose instead of ATen's generic strided
    copy -- and, in the backward, the gradient arrives channel-last again.  Any other layout / dtype: plain flatten."""
    if (x.dim() == 4 and x.is_cuda and x.dtype == torch.float32 and x.shape[0] <= 65535
            and x.permute(0, 2, 3, 1).is_contiguous() and not x.is_contiguous()):
        B, C, H, W = x.shape
        rows = x.permute(0, 2, 3, 1).reshape(B, H * W, C)
        return _TransposeBatched.apply(rows).reshape(B, C * H * W)
    return torch.flatten(x, start_dim=1)
