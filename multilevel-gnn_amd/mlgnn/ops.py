"""torch.autograd bindings of the CSR aggregation kernels (C ABI: include/mlgnn.h).

PyTorch owns memory and the autograd graph; every numeric step of the aggregation itself runs in
libmlgnn.so.  All functions require CUDA(HIP) fp32 tensors and raise otherwise -- no CPU path.
"""
import ctypes
import os

import torch

from . import _lib
from .graph import CSRGraph

MSG_IDENTITY, MSG_WEIGHTED, MSG_GEN = 0, 1, 2
EDGE_NONE, EDGE_RANK1, EDGE_FULL = 0, 1, 2
AGGR_SUM, AGGR_MEAN, AGGR_MAX, AGGR_SOFTMAX, AGGR_POWER = 0, 1, 2, 3, 4
DTYPE_F32, DTYPE_BF16 = 0, 1
_DTYPE_IDS = {torch.float32: DTYPE_F32, torch.bfloat16: DTYPE_BF16}
POW_LO, POW_HI = 1e-7, 1e1        # torch_message.py:69

# reference aggregator names (torch_message.py:14,27,45-82) -> kernel aggregator
AGGR_IDS = {"add": AGGR_SUM, "sum": AGGR_SUM, "mean": AGGR_MEAN, "max": AGGR_MAX,
            "softmax": AGGR_SOFTMAX, "softmax_sg": AGGR_SOFTMAX, "softmax_sum": AGGR_SOFTMAX,
            "power": AGGR_POWER, "power_sum": AGGR_POWER}


def _stream():
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """HIP-event stopwatch around individual C-ABI launches (events are recorded on the stream the
    kernel is launched on -- torch's current stream).  ``bench.py`` installs one to report the
    roofline fraction of the aggregation kernels from live measurements."""

    def __init__(self):
        self.records = []           # (name, start_event, end_event, algorithmic_bytes)

    def start(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def stop(self, name, start, nbytes):
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        self.records.append((name, start, end, nbytes))

    def summary(self):
        """-> {name: dict(launches, avg_ms, bytes)}; call after torch.cuda.synchronize()."""
        out = {}
        for name, s, e, nbytes in self.records:
            d = out.setdefault(name, dict(launches=0, total_ms=0.0, bytes=nbytes))
            d["launches"] += 1
            d["total_ms"] += s.elapsed_time(e)
        for d in out.values():
            d["avg_ms"] = d["total_ms"] / d["launches"]
        return out


KERNEL_TIMER = None
_AGGR_NAMES = {AGGR_SUM: "sum", AGGR_MEAN: "mean", AGGR_MAX: "max", AGGR_SOFTMAX: "softmax", AGGR_POWER: "power"}
_EDGE_NAMES = {EDGE_NONE: "noedge", EDGE_RANK1: "rank1", EDGE_FULL: "full"}


def _edge_name(edge_mode, rank):
    return "rank%d" % rank if edge_mode == EDGE_RANK1 else _EDGE_NAMES[edge_mode]


def algorithmic_bytes(N, E, d, aggr_id, edge_mode, backward=False, learn_t=False, weighted=False, gen=True, s=4,
                      rank=1):
    """Edge-gather byte count of one launch, no cache credit.
    ``s`` = bytes per activation element (4 fp32, 2 bf16); lse / argmax side arrays are 4-byte."""
    rows = E * d * s                                  # one gathered activation row per edge
    idx = E * 4 + (N + 1) * 4                         # col + rowptr
    scalar = E * 4 * rank if edge_mode == EDGE_RANK1 else (E * 4 if weighted else 0)
    full = (E * d * s + E * 4) if edge_mode == EDGE_FULL else 0
    if not backward:
        extra = N * d * 4 if aggr_id == AGGR_MAX else 0          # argmax write
        return rows + idx + scalar + full + N * d * s + extra
    gathers = rows                                                   # grad_out rows
    prepass = 0
    if aggr_id == AGGR_SOFTMAX and learn_t:
        gathers += E * d * 4 + rows                                   # lse + out rows for the recompute
    elif aggr_id == AGGR_SOFTMAX:
        # one-row path: the gathered row is gt = go * 2^(-lse); the prepass streams go + lse in and gt out once
        # per node
        prepass = N * d * s + N * d * 4 + N * d * s + (N + 1) * 4
    if aggr_id == AGGR_MAX:
        gathers += E * d * 4 + E * 4                                  # argmax rows + pos_t
    if edge_mode == EDGE_FULL:
        full += E * d * s                                             # grad_efull write
    own = (N * d * s if gen else 0) + N * d * s                       # x_j read + grad_x write
    return gathers + idx + scalar + full + own + prepass


def _dev_f32(t, what):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: libmlgnn has no CPU path" % what)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (what, t.dtype))
    return t.contiguous()


def _dev_act(t, what, like=None):
    """Activation tensor: fp32 or bf16 storage (arithmetic is fp32 in the kernels either way)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: libmlgnn has no CPU path" % what)
    if t.dtype not in _DTYPE_IDS:
        raise TypeError("%s must be float32 or bfloat16, got %s" % (what, t.dtype))
    if like is not None and t.dtype != like.dtype:
        t = t.to(like.dtype)
    return t.contiguous()


def tag_row_max(t, row_max):
    """Attach ``max |row|`` ([N] fp32, written by the kernel that produced ``t``) to a 2-D tensor; the tall GEMM
    reads it instead of streaming its operand twice.  Tied to the tensor's version: an in-place edit voids it."""
    t._mlgnn_row_max = (row_max, t._version)
    return t


SHIFT_STATS = {"given": 0, "computed": 0}     # softmax backwards whose rescaled cotangent came from the producer / a pre-pass


def tag_softmax_lse(out, lse, rowptr):
    """Mark ``out`` as the result of a softmax aggregation with log-sum-exp ``lse`` [N,d] over the CSR ``rowptr``: the
    Linear that consumes it can then emit, from its input-gradient GEMM, the rescaled cotangent this aggregation's
    backward gathers (:func:`tag_shifted`)."""
    out._mlgnn_lse = (lse, rowptr, out._version)
    return out


def softmax_lse_of(t):
    tag = getattr(t, "_mlgnn_lse", None)
    if tag is not None and tag[2] == t._version and tag[0].shape == t.shape:
        return tag[0], tag[1]
    return None


def tag_shifted(grad, gt, flag, lse):
    """Attach ``gt = grad * 2^(-lse)`` and its overflow flag to a cotangent on its way to the aggregation's backward."""
    grad._mlgnn_gt = (gt, flag, lse.data_ptr(), grad._version)
    return grad


def shifted_of(grad, lse):
    tag = getattr(grad, "_mlgnn_gt", None)
    if (tag is not None and tag[3] == grad._version and tag[2] == lse.data_ptr() and tag[0].shape == grad.shape
            and grad.is_contiguous() and tag[0].dtype == grad.dtype):
        return tag[0], tag[1]
    return None


class PostLN:
    """Side channel between the three nodes around a res+ block's pre-conv ``y = relu?(LayerNorm(h))``
    (deepergcn.py:236-241) when ``y`` was written by the previous conv's last GEMM (:class:`mlgnn.dense._FusedMLP2`):

    * the aggregation that consumes ``y`` takes the finished ``d loss / d y`` rows through the LayerNorm's backward in
      its own row epilogue (``mlgnn_csr_aggregate_bwd_ln``), returns NO gradient for ``y`` and leaves
      ``folded = (d loss / d h, d gamma, d beta, row maxima)`` here;
    * the op that adds ``h`` as its residual (the same block's MLP) leaves the gradient of that identity branch in
      ``extra`` instead of returning it, so that the epilogue above adds it in the same pass (``extra_used``);
    * the producer of ``(h, y)`` picks both up in its backward -- and still runs the separate LayerNorm backward on
      whatever gradient reaches ``y`` from other consumers.
    Every field is consumed (reset) by the backward that reads it."""

    def __init__(self, h, mean, rstd, gamma, beta, relu):
        self.h, self.mean, self.rstd, self.gamma, self.beta, self.relu = h, mean, rstd, gamma, beta, bool(relu)
        self.extra = None
        self.extra_used = False
        self.folded = None


# Off by default.  Measured at BASELINE configs[1] (same-box A/B, bench.py): the epilogue costs the aggregation backward
# 0.13-0.27 ms per launch (per row / per pair of rows, operands prefetched or not, non-temporal or not) against the
# 0.21 ms LayerNorm-backward launch it removes -- that kernel is bound by instruction issue and by the L2 hit rate of
# its gather, and the epilogue adds to both; the step moved by -0.1 ... +0.1 ms.  MLGNN_LN_FOLD=1 turns it on.
LN_FOLD = os.environ.get("MLGNN_LN_FOLD", "0") == "1"
LN_FOLD_STATS = {"folded": 0, "separate": 0}


if os.environ.get("MLGNN_PRINT_STATS", "0") == "1":          # development: which paths a run took
    import atexit
    import sys
    atexit.register(lambda: print("mlgnn stats: ln_fold %r" % (LN_FOLD_STATS,), file=sys.stderr))


def tag_post_ln(y, h, tag):
    y._mlgnn_post_ln = tag
    h._mlgnn_post_ln_of = tag
    return y


_PARAM_EPOCH = [0]          # bumped after every step of ANY torch.optim.Optimizer (global post-step hook below)


def _after_optimizer_step(optimizer, args, kwargs):
    _PARAM_EPOCH[0] += 1
    for group in optimizer.param_groups:
        for p in group["params"]:
            p._mlgnn_stepped = True


try:                                                        # (public since torch 2.0)
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_hook
    _reg_post_hook(_after_optimizer_step)
except ImportError:                                         # pragma: no cover
    pass


def invalidate_param_cache():
    """Call after editing parameters behind autograd's back outside an optimizer step (``p.data.copy_``, an EMA swap that
    keeps the storage): drops every cached fp32 copy at its next use."""
    _PARAM_EPOCH[0] += 1


def f32_cached(t):
    """``t`` as a contiguous fp32 tensor for a kernel argument (LayerNorm gamma / beta, a bias: the kernels read their
    [d]-sized parameters in fp32).  A non-fp32 tensor is cast per call -- always correct -- unless it is a parameter some
    ``torch.optim.Optimizer`` has stepped: its copy is then kept until the next optimizer step of the process (a global
    post-step hook counts them: updates through ``p.data.copy_`` inside an optimizer, as the reference's utils/optim.py
    does, are seen although they do not bump the version counter), a version bump, or a change of storage address /
    device, so a bf16 model casts each parameter once per step instead of once per use (~20 tiny launches per layer at
    BASELINE configs[4]).  Edits through ``.data`` between optimizer steps: :func:`invalidate_param_cache`.
    Only for use inside autograd Functions (the copy is detached)."""
    if t.dtype == torch.float32:
        return t.contiguous()
    if not getattr(t, "_mlgnn_stepped", False):
        return t.detach().float().contiguous()
    key = (_PARAM_EPOCH[0], t._version, t.data_ptr(), t.device)
    tag = getattr(t, "_mlgnn_f32", None)
    if tag is not None and tag[0] == key:
        return tag[1]
    c = t.detach().float().contiguous()
    t._mlgnn_f32 = (key, c)
    return c


def row_max_of(t):
    tag = getattr(t, "_mlgnn_row_max", None)
    if tag is not None and tag[1] == t._version and tag[0].shape[0] == t.shape[0]:
        return tag[0]
    return None


class LowRankEdge:
    """Edge embedding kept in factored form: ``e_ij = weight @ a_ij + bias``.

    ``a`` holds the raw edge attributes ``[E, r]`` in COO order (r <= 8), ``weight`` is ``[H, r]`` and
    ``bias`` ``[H]`` -- the parameters of ``DeeperGCN.edge_encoder = Linear(7 or 1, H)``
    (deepergcn.py:87-90,213).  The factorisation survives ``GENConv.edge_encoder = Linear(H, d)``
    (torch_vertex.py:68,77), so the ``[E, d]`` embedding and its 2*E*H*d FLOP GEMM per layer never
    have to exist: the aggregation kernels rebuild ``e_ij`` from r scalars per edge.
    """
    MAX_RANK = 8

    def __init__(self, a, weight, bias):
        if a.dim() == 1:
            a = a[:, None]
        if weight.dim() == 1:
            weight = weight[:, None]
        if a.shape[1] != weight.shape[1]:
            raise ValueError("edge attributes are [E, %d] but the encoder takes %d columns" % (a.shape[1], weight.shape[1]))
        if not 1 <= a.shape[1] <= self.MAX_RANK:
            raise ValueError("factored edge term supports 1..%d attribute columns, got %d" % (self.MAX_RANK, a.shape[1]))
        self.a = a
        self.weight = weight
        self.bias = bias

    @property
    def rank(self):
        return self.a.shape[1]

    def through_linear(self, W, b):
        """Compose with ``Linear``: ``W (U a + c) + b = (W U) a + (W c + b)``.  The composed factors are fp32
        whatever the model's storage type (the kernels keep them in fp32 registers anyway; and a [d, H] x [H, r]
        product in bf16 goes through the GEMM library's per-call heuristics: 2-5 ms of host time each)."""
        Wf = W.float()
        cb = torch.mv(Wf, self.bias.float())
        return LowRankEdge(self.a, torch.mm(Wf, self.weight.float()), cb if b is None else cb + b.float())

    def dense(self):
        return torch.addmm(self.bias, self.a.to(self.weight.dtype), self.weight.t())


class RankOneEdge(LowRankEdge):
    """``e_ij = a_ij * weight + bias`` with a scalar attribute ``a [E]`` and ``weight``/``bias`` ``[H]``
    (``Linear(1, H)``: the ``use_column`` configurations)."""

    def __init__(self, a, weight, bias):
        super().__init__(a.reshape(-1, 1), weight.reshape(-1, 1), bias)


def padded_rank(r):
    """Kernel instantiations exist for 1, 2, 4 and 8 scalars per edge; attributes are zero padded."""
    return 1 if r <= 1 else 2 if r <= 2 else 4 if r <= 4 else 8


class _EdgeTypeEmbedding(torch.autograd.Function):
    """``table[idx]`` for the per-edge type embedding of DeeperGCN (deepergcn.py:103-104,213) with the gradient
    ``grad_table[t] = sum_{e: idx[e] = t} grad_e[e]`` on ``csrc/embedding.hip``: the edge ids are sorted by type once
    (stable), then every table row gathers and sums its cotangent rows in a fixed order."""

    @staticmethod
    def forward(ctx, table, idx):
        ctx.save_for_backward(idx)
        ctx.rows = table.shape[0]
        return table.index_select(0, idx)

    @staticmethod
    def backward(ctx, ge):
        (idx,) = ctx.saved_tensors
        T, d = ctx.rows, ge.shape[1]
        ge = ge.contiguous()
        order = torch.sort(idx, stable=True)[1].to(torch.int32)
        rowptr = torch.zeros(T + 1, dtype=torch.int64, device=idx.device)
        torch.cumsum(torch.bincount(idx, minlength=T), 0, out=rowptr[1:])
        rowptr = rowptr.to(torch.int32)
        out = torch.empty((T, d), dtype=torch.float32, device=ge.device)
        rc = _lib.lib.mlgnn_embedding_bwd(ge.data_ptr(), order.data_ptr(), rowptr.data_ptr(), out.data_ptr(), T, d,
                                          DTYPE_F32, _stream())
        _lib.check(rc, "mlgnn_embedding_bwd")
        return out, None


def edge_type_embedding(table, idx):
    """``nn.Embedding`` forward for a 1-D long ``idx`` with a deterministic, gather-speed backward; ATen's
    ``F.embedding`` for anything the kernel does not cover (non-fp32, d % 4 != 0, CPU tensors)."""
    if (table.is_cuda and table.dtype == torch.float32 and table.dim() == 2 and table.shape[1] % 4 == 0
            and idx.dim() == 1 and idx.dtype == torch.long and idx.numel() < 2 ** 31):
        return _EdgeTypeEmbedding.apply(table, idx)
    return torch.nn.functional.embedding(idx, table)


# max + TableEdge: the table gradient summed inside the backward kernel (fixed-point atomics) instead of through an [E, d]
# per-edge gradient.  Saves that buffer (5.2 GB for a BASELINE configs[1] batch) but measured 3 % SLOWER per step with
# the reference's default DeeperGCN flags (22.6 vs 21.9 ms, same box): 82 M sparse 8-byte atomics per layer execute at
# the memory side, one 64-byte request each.  Off by default.
TABLE_DIRECT = os.environ.get("MLGNN_TABLE_DIRECT", "0") == "1"
# max + TableEdge, the default since round 4: the table gradient from the DESTINATION side -- the forward's argmax names
# the winner of every (node, channel), so the gradient is one streaming pass over grad_out and argmax
# (mlgnn_max_table_grad) and the aggregation backward writes nothing per edge.  MLGNN_TABLE_DEST=0: the per-edge buffer.
TABLE_DEST = os.environ.get("MLGNN_TABLE_DEST", "1") == "1"
# max aggregator, the backward from compact winner lists (csrc/max_sparse.hip): on by default where it applies
SPARSE_MAX = os.environ.get("MLGNN_SPARSE_MAX", "1") == "1"
SPARSE_MAX_STATS = {"calls": 0, "table": 0}
TABLE_SLOTS = os.environ.get("MLGNN_TABLE_SLOTS", "1") == "1"      # A/B: the by-type pass reads one-byte winner slots
TABLE_DEST_STATS = {"calls": 0, "streamed": 0, "by_type": 0}


class _GradSink:
    """Accumulation buffer shared by the aggregation layers that consume one dense edge embedding."""

    def __init__(self):
        self.buf = None
        self.graph = None          # TableEdge: the graph whose by-source edge order the rows of buf follow
        # TableEdge under the max aggregator: the table's gradient is summed directly (fixed-point accumulator,
        # csrc/embedding.hip) -- `total` collects the layers of one backward
        self.fix = None
        self.total = None


class _EdgeFanout(torch.autograd.Function):
    """Identity on a dense ``[E, d]`` edge embedding that several aggregation layers read (the reference hands the
    same ``edge_emb`` to every GENConv, deepergcn.py:232-281).  Its consumers add their edge gradients into ONE buffer
    inside their backward kernels (``accumulate_efull``) and report no gradient of their own; this node hands the
    buffer on.  Autograd would otherwise sum L tensors of E*d elements with L-1 separate passes (measured at BASELINE
    configs[1] size, 3 layers: 5 ms of a 33 ms step)."""

    @staticmethod
    def forward(ctx, e):
        ctx.sink = _GradSink()
        ctx.set_materialize_grads(False)
        return e.view_as(e)

    @staticmethod
    def backward(ctx, g):
        total, ctx.sink.buf = ctx.sink.buf, None
        if g is not None:                       # a consumer outside the aggregation kernels contributed as well
            total = g if total is None else total + g
        return total


class _TableFanout(torch.autograd.Function):
    """Identity on the ``[T, d]`` table of a :class:`TableEdge`.  The aggregation layers that read it produce their
    edge gradient per EDGE (``[E, d]``, accumulated across layers in ``sink.buf`` inside the backward kernels); this
    node reduces it to the table once, ``grad_table[t] = sum_{e: idx[e] = t}`` (``csrc/embedding.hip``)."""

    @staticmethod
    def forward(ctx, table, owner):
        ctx.sink, ctx.owner = _GradSink(), owner
        ctx.set_materialize_grads(False)
        return table.view_as(table)

    @staticmethod
    def backward(ctx, g):
        per_edge, ctx.sink.buf = ctx.sink.buf, None
        direct, ctx.sink.total = ctx.sink.total, None
        total = None
        if per_edge is not None:
            order, rowptr = ctx.owner.sorted_by_type(ctx.sink.graph)
            T, d = ctx.owner.table_rows, per_edge.shape[1]
            total = torch.empty((T, d), dtype=torch.float32, device=per_edge.device)
            rc = _lib.lib.mlgnn_embedding_bwd(per_edge.data_ptr(), order.data_ptr(), rowptr.data_ptr(),
                                              total.data_ptr(), T, d, DTYPE_F32, _stream())
            _lib.check(rc, "mlgnn_embedding_bwd")
        if direct is not None:
            total = direct if total is None else total + direct
        if g is not None:
            total = g if total is None else total + g
        return total, None


class TableEdge:
    """Edge embedding read through a table: ``e_ij = table[idx_ij]`` -- ``table`` ``[T, d]``, ``idx`` ``[E]`` long in
    COO order.  This is DeeperGCN's default edge term (``global_edge='onehot'``: ``nn.Embedding(pathway_edge_num, H)``
    applied to every edge, deepergcn.py:103-104,189-190,213): the ``[E, H]`` embedding is never materialised, the
    aggregation kernels read the (cache-resident) table row of every edge, and the per-edge gradient the backward
    kernels produce is reduced to the table once for all layers that share it.  ``through_linear`` composes with a
    per-layer ``Linear(H, d)`` edge encoder (torch_vertex.py:68,77) as a ``[T, H] x [H, d]`` product."""

    def __init__(self, table, idx, source=None):
        """``source``: the tensor ``idx`` was derived from (the batch's ``edge_attr``).  The index arrays derived from
        ``idx`` for a graph (rows in by-destination / by-source order, the sort by table row) are then kept ON THE GRAPH
        and reused while the same ``source`` (same elements, same version) comes back with the same graph -- a training
        loop over one graph (BASELINE configs[4]) or a reused batch pays the gathers and the sort once, not per step."""
        if table.dim() != 2 or idx.dim() != 1 or idx.dtype != torch.long:
            raise ValueError("TableEdge takes table [T, d] and idx [E] (long)")
        self.idx = idx
        self.source = source
        self.table_rows = table.shape[0]
        self.sink = None
        if table.requires_grad and torch.is_grad_enabled() and table.is_cuda and table.dtype == torch.float32:
            table = _TableFanout.apply(table, self)
            self.sink = table.grad_fn.sink
        self.table = table
        self._by_graph = {}
        self._sorted = None

    def through_linear(self, W, b):
        return TableEdge(torch.nn.functional.linear(self.table, W, b), self.idx, self.source)

    def _graph_cache(self, graph):
        """The dict on ``graph`` that holds what this term derived from (``source``, graph): valid while ``source`` is the
        same view at the same version (the entry holds the tensor, so its storage cannot be recycled under it)."""
        from .graph import _same_view
        src = self.source
        if src is None:
            return None
        ent = getattr(graph, "_table_edge_cache", None)
        if (ent is None or ent[1] != src._version or ent[2] != self.table_rows or not _same_view(ent[0], src)):
            ent = (src, src._version, self.table_rows, {})
            graph._table_edge_cache = ent
        return ent[3]

    def dense(self):
        return self.table.index_select(0, self.idx)

    def rows_for(self, graph):
        """``(table row per edge in by-destination order, in by-source order, arange(E))`` (int32), cached per graph.
        The per-edge gradient is laid out in BY-SOURCE order -- the order the backward kernel walks the edges in -- so
        that it is written (and, from the second layer on, re-read) as a stream instead of as scattered rows."""
        key = id(graph)
        if key not in self._by_graph:
            shared = self._graph_cache(graph)
            if shared is not None and "rows" in shared:
                self._by_graph[key] = shared["rows"]
                return self._by_graph[key]
            i32 = self.idx.to(torch.int32)
            self._by_graph[key] = (i32[graph.eid.long()].contiguous(), i32[graph.eid_t.long()].contiguous(),
                                   torch.arange(self.idx.numel(), dtype=torch.int32, device=self.idx.device))
            if shared is not None:
                shared["rows"] = self._by_graph[key]
        return self._by_graph[key]

    def sorted_by_type(self, graph):
        """By-source edge positions sorted (stably) by table row, and the row pointer over them (int32)."""
        key = id(graph)
        if self._sorted is None or self._sorted[0] != key:
            shared = self._graph_cache(graph)
            if shared is not None and "sorted" in shared:
                self._sorted = (key,) + shared["sorted"]
                return self._sorted[1], self._sorted[2]
            by_src = self.rows_for(graph)[1].long()
            order = torch.sort(by_src, stable=True)[1].to(torch.int32)
            rowptr = torch.zeros(self.table_rows + 1, dtype=torch.int64, device=self.idx.device)
            torch.cumsum(torch.bincount(by_src, minlength=self.table_rows), 0, out=rowptr[1:])
            self._sorted = (key, order, rowptr.to(torch.int32))
            if shared is not None:
                shared["sorted"] = (self._sorted[1], self._sorted[2])
        return self._sorted[1], self._sorted[2]


    def winners_by_type(self, graph):
        """``(by-destination edge positions sorted (stably) by table row, the destination node of each, row pointer,
        position of each edge inside its destination row)`` (int32) -- what mlgnn_max_table_grad_by_type walks: destination order inside a table row, so the rows gathered for
        it run through the batch graph by graph.  Derived once per (graph, source), like the other index arrays."""
        shared = self._graph_cache(graph)
        store = shared if shared is not None else self.__dict__.setdefault("_winners", {})
        key = "winners" if shared is not None else id(graph)
        if key not in store:
            by_dst = self.rows_for(graph)[0].long()
            order = torch.sort(by_dst, stable=True)[1]
            rowptr = torch.zeros(self.table_rows + 1, dtype=torch.int64, device=self.idx.device)
            torch.cumsum(torch.bincount(by_dst, minlength=self.table_rows), 0, out=rowptr[1:])
            rp = graph.rowptr.long()
            dst = torch.searchsorted(rp, order, right=True) - 1
            store[key] = (order.to(torch.int32), dst.to(torch.int32), rowptr.to(torch.int32), (order - rp[dst]).to(torch.int32))
        return store[key]


def share_edge_gradient(e):
    """Mark a dense edge embedding as shared by several :func:`gen_aggregate` calls (see :class:`_EdgeFanout`)."""
    if not (torch.is_tensor(e) and e.dim() == 2 and e.requires_grad and torch.is_grad_enabled()):
        return e
    out = _EdgeFanout.apply(e)
    out._mlgnn_grad_sink = out.grad_fn.sink
    return out


class _GenAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eu, ev, efull, t_par, p_par, graph, ew_pair, aggr_id, t, p, eps, learn_t, learn_p, add_root,
                table_edge=None):
        post_ln = getattr(x, "_mlgnn_post_ln", None)
        x = _dev_act(x, "x")
        dtype_id = _DTYPE_IDS[x.dtype]
        N, d = x.shape
        # x = relu?(LayerNorm(h)) written by the previous conv's GEMM: its backward can run in this op's row epilogue
        ctx.post_ln = post_ln if (LN_FOLD and post_ln is not None and add_root and not learn_t and not learn_p
                                  and x.dtype == torch.float32 and d in (16, 32, 64, 128, 256)
                                  and aggr_id != AGGR_POWER and post_ln.h.shape == x.shape) else None
        if graph.num_nodes != N:
            raise ValueError("graph/feature size mismatch")
        edge_mode = EDGE_RANK1 if eu is not None else (EDGE_FULL if efull is not None else EDGE_NONE)
        ctx.uv_dtype = eu.dtype if eu is not None else None
        ctx.grad_sink = getattr(efull, "_mlgnn_grad_sink", None) if efull is not None else None
        ctx.table_edge = table_edge
        eid_fwd = graph.eid
        if table_edge is not None:                      # efull is the table; every edge names its row
            eid_fwd = table_edge.rows_for(graph)[0]
            ctx.grad_sink = table_edge.sink
        eu = _dev_f32(eu.float(), "eu") if eu is not None else None       # edge vectors stay fp32 in the kernel
        ev = _dev_f32(ev.float(), "ev") if ev is not None else None
        efull = _dev_act(efull, "efull", like=x)
        if efull is not None and table_edge is None and tuple(efull.shape) != (graph.num_edges, d):
            raise ValueError("edge embedding must be [E, d]")
        if table_edge is not None and (efull.shape[1] != d or table_edge.idx.numel() != graph.num_edges):
            raise ValueError("table edge term must be table [T, d] with one row index per edge")
        rank = 0
        if eu is not None:
            if eu.dim() != 2 or eu.shape[1] != d or ev.numel() != d:
                raise ValueError("factored edge term must be eu [r, d], ev [d]")
            rank = padded_rank(eu.shape[0])
            ctx.uv_rows = eu.shape[0]
            if eu.shape[0] != rank:                                   # pad with zero rows to the kernel's rank
                eu = torch.cat([eu, eu.new_zeros(rank - eu.shape[0], d)], dim=0)
            if ew_pair[0].shape != (graph.num_edges, rank):
                raise ValueError("edge attribute table must be [E, %d]" % rank)
        out = torch.empty_like(x)
        want_bwd = any(ctx.needs_input_grad)
        f32 = dict(dtype=torch.float32, device=x.device)
        aux = torch.empty((N, d), **f32) if (aggr_id in (AGGR_SOFTMAX, AGGR_POWER) and want_bwd) else None
        aux2 = torch.empty((N, d), **f32) if ((aggr_id == AGGR_SOFTMAX and learn_t) or
                                              (aggr_id == AGGR_POWER and learn_p)) else None
        argmax = torch.empty((N, d), dtype=torch.int32, device=x.device) if aggr_id == AGGR_MAX else None
        ew = ew_pair[0] if ew_pair is not None else None
        t_dev = t_par if (learn_t and t_par is not None) else None
        p_dev = p_par if (learn_p and p_par is not None) else None
        if (t_dev is not None and t_dev.dtype != torch.float32) or (p_dev is not None and p_dev.dtype != torch.float32):
            raise TypeError("learnable t / p must stay float32 (keep them out of a bf16 cast)")
        timer = KERNEL_TIMER
        t0 = timer.start() if timer is not None else None
        # max |row| of the result rides along for the Linear that consumes it (fp32, one channel chunk)
        rowmax = torch.empty(N, **f32) if (x.dtype == torch.float32 and d in (4, 8, 16, 32, 64, 128, 256)) else None
        hub, hub_keep = graph.hub_arg("dst", d)
        if hub is not None and aggr_id in (AGGR_SOFTMAX, AGGR_POWER) and aux is None:
            aux = torch.empty((N, d), **f32)                  # the chunks of a long row are combined through their lse
        rc = _lib.lib.mlgnn_csr_aggregate_fwd(
            x.data_ptr(), graph.rowptr.data_ptr(), graph.col.data_ptr(), _lib.ptr(ew), _lib.ptr(eu), _lib.ptr(ev),
            _lib.ptr(efull), eid_fwd.data_ptr(), out.data_ptr(), _lib.ptr(aux), _lib.ptr(aux2),
            _lib.ptr(argmax), _lib.ptr(rowmax), N, d, dtype_id, MSG_GEN, edge_mode, rank, aggr_id, float(t), float(p),
            _lib.ptr(t_dev), _lib.ptr(p_dev), float(eps), int(add_root), hub, _stream())
        _lib.check(rc, "mlgnn_csr_aggregate_fwd")
        del hub_keep
        if timer is not None:
            timer.stop("csr_aggregate_fwd/%s/%s" % (_AGGR_NAMES[aggr_id], _edge_name(edge_mode, rank)), t0,
                       algorithmic_bytes(N, graph.num_edges, d, aggr_id, edge_mode, s=x.element_size(),
                                         rank=max(rank, 1)))
        ctx.graph, ctx.ew_pair = graph, ew_pair
        ctx.cfg = (aggr_id, edge_mode, rank, float(t), float(p), float(eps), bool(learn_t), bool(learn_p), bool(add_root))
        ctx.save_for_backward(x, out, aux, aux2, argmax, eu, ev, efull, t_dev, p_dev)
        ctx.rowmax = rowmax
        # softmax without a learnable temperature: the backward gathers go * 2^(-lse); let the consumer of `out` know
        ctx.lse_for_shift = aux if (aggr_id == AGGR_SOFTMAX and not learn_t and aux is not None) else None
        return out

    @staticmethod
    def backward(ctx, go):
        x, out, aux, aux2, argmax, eu, ev, efull, t_dev, p_dev = ctx.saved_tensors
        aggr_id, edge_mode, rank, t, p, eps, learn_t, learn_p, add_root = ctx.cfg
        g = ctx.graph
        N, d = x.shape
        # the producer of grad_out may already have written the rescaled cotangent the softmax backward gathers
        # (mlgnn.dense: the input-gradient GEMM of the Linear behind this aggregation, csrc/tallgemm.hip SHIFT)
        shifted = shifted_of(go, aux) if (aggr_id == AGGR_SOFTMAX and not learn_t and aux is not None) else None
        go = _dev_act(go, "grad_out", like=x)
        dtype_id = _DTYPE_IDS[x.dtype]
        grad_t = grad_p = None
        go_k = go
        if aggr_id == AGGR_POWER:
            # the cotangent through the outer power and the mean (and d loss / dp): one streaming pass (csrc/power.hip)
            go_k = torch.empty_like(go)
            gp = torch.empty(1, dtype=torch.float32, device=x.device) if learn_p else None
            pw_n = int(_lib.lib.mlgnn_power_bwd_prologue_workspace_floats())
            pw_ws = torch.empty(pw_n, dtype=torch.float32, device=x.device) if learn_p else None
            rc = _lib.lib.mlgnn_power_bwd_prologue(go.data_ptr(), aux.data_ptr(), g.rowptr.data_ptr(), _lib.ptr(out if learn_p else None),
                                                   _lib.ptr(aux2 if learn_p else None), float(p), _lib.ptr(p_dev), go_k.data_ptr(),
                                                   _lib.ptr(gp), _lib.ptr(pw_ws), pw_n, N, d, dtype_id, _stream())
            _lib.check(rc, "mlgnn_power_bwd_prologue")
            if learn_p:
                grad_p = gp.to(p_dev.dtype)
        if aggr_id == AGGR_SOFTMAX and learn_t:
            grad_t = (go.float() * (aux2 - out.float() * out.float())).sum().reshape(1).to(t_dev.dtype)
        gx = torch.empty_like(x)
        ge, ge_accumulate, sink, te = None, 0, ctx.grad_sink, ctx.table_edge
        eid_t, geid_t = g.eid_t, None
        if te is not None:                                   # read the table row, write the edge's own gradient row
            _, eid_t, geid_t = te.rows_for(g)                # (gradient rows in by-source order: a streamed write)
            if sink is not None:
                if sink.graph is not None and sink.graph is not g:
                    raise RuntimeError("a TableEdge is tied to one graph (one batch)")
                sink.graph = g
        # max over a table edge term: only the winning edge of (i, c) has a gradient -- it goes straight to the table's
        # fixed-point accumulator inside the kernel (no [E, d] gradient written, re-read and reduced)
        fix_table = (TABLE_DIRECT and te is not None and sink is not None and aggr_id == AGGR_MAX and edge_mode == EDGE_FULL
                     and x.dtype == torch.float32 and d % 4 == 0 and ctx.post_ln is None)
        sparse = (SPARSE_MAX and not fix_table and aggr_id == AGGR_MAX and x.dtype == torch.float32 and argmax is not None
                  and ctx.post_ln is None and g.num_edges > 0 and go_k.data_ptr() % 16 == 0 and argmax.data_ptr() % 16 == 0
                  and (edge_mode == EDGE_NONE or (edge_mode == EDGE_FULL and te is not None and TABLE_DEST))
                  and int(_lib.lib.mlgnn_max_sparse_records(N, d, g.num_edges)) > 0 and g.known_short_rows())
        if sparse:
            # every (node, channel) has ONE winning edge: its cotangent goes to that edge's source -- and to the table row
            # the edge reads -- through compact per-edge runs of (value, channel) pairs instead of whole gathered rows
            dev = x.device
            recs = torch.empty((int(_lib.lib.mlgnn_max_sparse_records(N, d, g.num_edges)), 2), dtype=torch.int32, device=dev)
            meta = torch.empty((g.num_edges, 2), dtype=torch.int32, device=dev)
            rc = _lib.lib.mlgnn_max_winners(go_k.data_ptr(), argmax.data_ptr(), g.rowptr.data_ptr(), recs.data_ptr(),
                                            meta.data_ptr(), N, d, _stream())
            _lib.check(rc, "mlgnn_max_winners")
            rc = _lib.lib.mlgnn_max_sparse_bwd(recs.data_ptr(), meta.data_ptr(), g.rowptr_t.data_ptr(), g.pos_t.data_ptr(),
                                               go_k.data_ptr() if add_root else None, gx.data_ptr(), N, d, _stream())
            _lib.check(rc, "mlgnn_max_sparse_bwd")
            SPARSE_MAX_STATS["calls"] += 1
            if te is not None and sink is not None:
                T = te.table_rows
                first = sink.total is None
                if first:
                    sink.total = torch.empty((T, d), dtype=torch.float32, device=dev)
                if bool(_lib.lib.mlgnn_max_table_grad_supported(N, d, T)):
                    rows_dst = te.rows_for(g)[0]              # a few table rows: the streaming pass with LDS partial tables
                    mt_n = int(_lib.lib.mlgnn_max_table_grad_workspace_floats(N, d, T))
                    mt_ws = torch.empty(mt_n, dtype=torch.float32, device=dev)
                    rc = _lib.lib.mlgnn_max_table_grad(go_k.data_ptr(), argmax.data_ptr(), rows_dst.data_ptr(),
                                                       sink.total.data_ptr(), mt_ws.data_ptr(), mt_n, N, d, T,
                                                       0 if first else 1, _stream())
                    _lib.check(rc, "mlgnn_max_table_grad")
                    TABLE_DEST_STATS["streamed"] += 1
                else:
                    pos_s, _, rp_s, _ = te.winners_by_type(g)
                    rc = _lib.lib.mlgnn_max_sparse_table_grad(recs.data_ptr(), meta.data_ptr(), pos_s.data_ptr(), rp_s.data_ptr(),
                                                              sink.total.data_ptr(), N, d, T, 0 if first else 1, _stream())
                    _lib.check(rc, "mlgnn_max_sparse_table_grad")
                    SPARSE_MAX_STATS["table"] += 1
                TABLE_DEST_STATS["calls"] += 1
            return gx, None, None, None, grad_t, grad_p, None, None, None, None, None, None, None, None, None, None
        by_type_after = False
        dest_table = (not fix_table and TABLE_DEST and te is not None and sink is not None and aggr_id == AGGR_MAX
                      and edge_mode == EDGE_FULL and x.dtype == torch.float32 and argmax is not None and d % 4 == 0
                      and go_k.data_ptr() % 16 == 0 and argmax.data_ptr() % 16 == 0)
        if dest_table:
            T = te.table_rows
            first = sink.total is None
            if first:
                sink.total = torch.empty((T, d), dtype=torch.float32, device=x.device)
            if bool(_lib.lib.mlgnn_max_table_grad_supported(N, d, T)):
                # a few table rows: one streaming pass, per-workgroup partial tables in LDS
                rows_dst = te.rows_for(g)[0]
                mt_n = int(_lib.lib.mlgnn_max_table_grad_workspace_floats(N, d, T))
                mt_ws = torch.empty(mt_n, dtype=torch.float32, device=x.device)
                rc = _lib.lib.mlgnn_max_table_grad(go_k.data_ptr(), argmax.data_ptr(), rows_dst.data_ptr(),
                                                   sink.total.data_ptr(), mt_ws.data_ptr(), mt_n, N, d, T, 0 if first else 1,
                                                   _stream())
                _lib.check(rc, "mlgnn_max_table_grad")
                TABLE_DEST_STATS["streamed"] += 1
            else:
                by_type_after = True                         # (below, behind the backward: it reads that call's winner slots)
            TABLE_DEST_STATS["calls"] += 1
            ge, ge_accumulate, geid_t = None, 3, None
        elif fix_table:
            T = te.table_rows
            if sink.fix is None or sink.fix.numel() != int(_lib.lib.mlgnn_table_grad_bytes(T, d)):
                sink.fix = torch.zeros(int(_lib.lib.mlgnn_table_grad_bytes(T, d)), dtype=torch.uint8, device=x.device)
            _lib.check(_lib.lib.mlgnn_table_grad_begin(go_k.data_ptr(), N, d, sink.fix.data_ptr(), _stream()),
                       "mlgnn_table_grad_begin")
            ge, ge_accumulate, geid_t = sink.fix, 2, eid_t
        elif edge_mode == EDGE_FULL and (te is None or sink is not None):
            if sink is not None and sink.buf is not None:
                ge, ge_accumulate = sink.buf, 1              # add this layer's share to the layers that ran before
            else:
                ge = torch.empty_like(efull) if te is None else torch.empty((g.num_edges, d), dtype=efull.dtype,
                                                                            device=efull.device)
                if sink is not None:
                    sink.buf = ge
        elif edge_mode == EDGE_FULL:                         # a table without gradient: scratch row space
            ge = torch.empty((g.num_edges, d), dtype=efull.dtype, device=efull.device)
        guv = ws = None
        ws_n = int(_lib.lib.mlgnn_csr_aggregate_bwd_workspace_floats(N, d, dtype_id, rank,
                                                                     AGGR_SUM if shifted is not None else aggr_id,
                                                                     int(learn_t)))
        if ws_n < 0:
            _lib.check(ws_n, "mlgnn_csr_aggregate_bwd_workspace_floats")
        if ws_n > 0:
            ws = torch.empty(ws_n, dtype=torch.float32, device=x.device)
        if edge_mode == EDGE_RANK1:
            guv = torch.empty((rank + 1, d), dtype=torch.float32, device=x.device)
        ew_t = ctx.ew_pair[1] if ctx.ew_pair is not None else None
        timer = KERNEL_TIMER
        t0 = timer.start() if timer is not None else None
        hub, hub_keep = g.hub_arg("src", d)
        tag = ctx.post_ln if (hub is None and ctx.needs_input_grad[0]) else None
        if tag is not None:
            # d loss / d y goes through the LayerNorm backward of y = relu?(LayerNorm(h)) in the row epilogue; gx is then
            # d loss / d h (+ the identity-branch gradient the block's MLP left in tag.extra)
            f32 = dict(dtype=torch.float32, device=x.device)
            ln_n = int(_lib.lib.mlgnn_csr_aggregate_bwd_ln_workspace_floats(N, d))
            ln_ws, ggb, row_max = torch.empty(ln_n, **f32), torch.empty((2, d), **f32), torch.empty(N, **f32)
            extra = tag.extra if (tag.extra is not None and tag.extra.dtype == torch.float32
                                  and tag.extra.shape == x.shape and tag.extra.is_contiguous()) else None
            gamma, beta = f32_cached(tag.gamma), f32_cached(tag.beta)
            st = _lib.LnFoldStruct(tag.h.data_ptr(), tag.mean.data_ptr(), tag.rstd.data_ptr(), gamma.data_ptr(),
                                   beta.data_ptr(), _lib.ptr(extra), row_max.data_ptr(), ggb.data_ptr(), ln_ws.data_ptr(),
                                   ln_n, int(tag.relu))
            rc = _lib.lib.mlgnn_csr_aggregate_bwd_ln(
                go_k.data_ptr(), x.data_ptr(), out.data_ptr(), _lib.ptr(aux), _lib.ptr(argmax),
                g.rowptr_t.data_ptr(), g.col_t.data_ptr(), g.pos_t.data_ptr(), g.rowptr.data_ptr(),
                _lib.ptr(ew_t), _lib.ptr(eu), _lib.ptr(ev), _lib.ptr(efull), eid_t.data_ptr(), _lib.ptr(geid_t),
                gx.data_ptr(), _lib.ptr(ge), _lib.ptr(guv), _lib.ptr(ws), ws_n,
                N, d, dtype_id, MSG_GEN, edge_mode, rank, aggr_id, int(learn_t), t, p,
                _lib.ptr(t_dev), _lib.ptr(p_dev), eps, int(add_root), ge_accumulate, None,
                _lib.ptr(shifted[0]) if shifted else None, _lib.ptr(shifted[1]) if shifted else None,
                ctypes.byref(st), _stream())
            _lib.check(rc, "mlgnn_csr_aggregate_bwd_ln")
            tag.folded = (gx, ggb[0], ggb[1], row_max)
            tag.extra_used = extra is not None
            LN_FOLD_STATS["folded"] += 1
            gx = None                                        # nothing reaches y through autograd: see PostLN
        else:
            LN_FOLD_STATS["separate"] += int(ctx.post_ln is not None)
            rc = _lib.lib.mlgnn_csr_aggregate_bwd(
                go_k.data_ptr(), x.data_ptr(), out.data_ptr(), _lib.ptr(aux), _lib.ptr(argmax),
                g.rowptr_t.data_ptr(), g.col_t.data_ptr(), g.pos_t.data_ptr(), g.rowptr.data_ptr(),
                _lib.ptr(ew_t), _lib.ptr(eu), _lib.ptr(ev), _lib.ptr(efull), eid_t.data_ptr(), _lib.ptr(geid_t),
                gx.data_ptr(), _lib.ptr(ge), _lib.ptr(guv), _lib.ptr(ws), ws_n,
                N, d, dtype_id, MSG_GEN, edge_mode, rank, aggr_id, int(learn_t), t, p,
                _lib.ptr(t_dev), _lib.ptr(p_dev), eps, int(add_root), ge_accumulate, hub,
                _lib.ptr(shifted[0]) if shifted else None, _lib.ptr(shifted[1]) if shifted else None, _stream())
            _lib.check(rc, "mlgnn_csr_aggregate_bwd")
        del hub_keep
        if by_type_after:
            # one row per KEGG membership (tens of thousands): a wavefront per table row gathers its edges' winners
            pos_s, dst_s, rp_s, rel_s = te.winners_by_type(g)
            off = int(_lib.lib.mlgnn_csr_aggregate_bwd_slots_offset_floats(N, d, rank))
            slots = ws.data_ptr() + 4 * off if (ws is not None and off >= 0 and TABLE_SLOTS) else None
            rc = _lib.lib.mlgnn_max_table_grad_by_type(go_k.data_ptr(), argmax.data_ptr(), dst_s.data_ptr(), pos_s.data_ptr(),
                                                       rel_s.data_ptr(), rp_s.data_ptr(), slots, sink.total.data_ptr(), N, d,
                                                       te.table_rows, 0 if first else 1, _stream())
            _lib.check(rc, "mlgnn_max_table_grad_by_type")
            TABLE_DEST_STATS["by_type"] += 1
        if fix_table:
            first = sink.total is None
            if first:
                sink.total = torch.empty((te.table_rows, d), dtype=torch.float32, device=x.device)
            _lib.check(_lib.lib.mlgnn_table_grad_finish(sink.fix.data_ptr(), sink.total.data_ptr(), te.table_rows, d,
                                                        0 if first else 1, _stream()), "mlgnn_table_grad_finish")
        SHIFT_STATS["given" if shifted else "computed"] += int(aggr_id == AGGR_SOFTMAX and not learn_t)
        if sink is not None or te is not None:
            ge = None                                        # reported once, by the fan-out node of the shared term
        if timer is not None:
            timer.stop("csr_aggregate_bwd/%s/%s" % (_AGGR_NAMES[aggr_id], _edge_name(edge_mode, rank)), t0,
                       algorithmic_bytes(N, g.num_edges, d, aggr_id, edge_mode, backward=True, learn_t=learn_t,
                                         s=x.element_size(), rank=max(rank, 1)))
        geu = guv[:ctx.uv_rows].to(ctx.uv_dtype) if guv is not None else None
        gev = guv[rank].to(ctx.uv_dtype) if guv is not None else None
        return gx, geu, gev, ge, grad_t, grad_p, None, None, None, None, None, None, None, None, None, None


def gen_aggregate(x, graph, edge=None, aggr="softmax", t=1.0, p=1.0, eps=1e-7, learn_t=False, learn_p=False,
                  add_root=False):
    """``aggregate(relu(x_j + e_ij) + eps)`` over incoming edges -- GENConv.message + aggregate
    (torch_vertex.py:94-101, torch_message.py:44-85) in one kernel.

    ``edge``: ``None`` | :class:`LowRankEdge` / :class:`RankOneEdge` (already composed to width d) |
    :class:`TableEdge` (a row of a ``[T, d]`` table per edge) | ``[E, d]`` tensor (COO order).
    ``t``/``p``: float, or the 1-element parameter when ``learn_t``/``learn_p``.
    ``*_sum`` variants return the un-scaled value; the caller applies ``deg ** sigmoid(y)``.
    ``add_root``: return ``x + aggregate`` from the same pass (GENConv's ``h = x + m``); ignored
    (done as a separate add) when the aggregate itself is needed for a learnable ``t``/``p``.
    """
    if not isinstance(graph, CSRGraph):
        raise TypeError("graph must be a CSRGraph")
    aggr_id = AGGR_IDS[aggr]
    eu = ev = efull = ew_pair = table_edge = None
    if isinstance(edge, TableEdge):
        if not (x.is_cuda and edge.table.dtype == x.dtype == torch.float32):
            edge = edge.dense()                          # (bf16 / CPU: the materialised embedding)
        else:
            table_edge, efull = edge, edge.table
            edge = None
    if isinstance(edge, LowRankEdge):
        if edge.weight.shape[0] != x.shape[1]:
            raise ValueError("factored edge term has width %d, features have %d" % (edge.weight.shape[0], x.shape[1]))
        eu, ev = edge.weight.t(), edge.bias                           # [r, d], [d]
        ew_pair = graph.edge_table(edge.a, padded_rank(edge.rank))
    elif edge is not None and table_edge is None:
        efull = edge
    t_par = t if torch.is_tensor(t) else None
    p_par = p if torch.is_tensor(p) else None
    t_val = 1.0 if t_par is not None else t
    p_val = 1.0 if p_par is not None else p
    lt, lp = bool(learn_t) and t_par is not None, bool(learn_p) and p_par is not None
    fuse_root = bool(add_root) and not lt and not lp and aggr_id != AGGR_POWER
    out = _GenAggregate.apply(x, eu, ev, efull, t_par, p_par, graph, ew_pair, aggr_id, t_val, p_val, eps,
                              lt, lp, fuse_root, table_edge)
    if add_root and not fuse_root:
        return out + x
    rm = getattr(out.grad_fn, "rowmax", None) if out.grad_fn is not None else None
    if rm is not None:
        tag_row_max(out, rm)
    lse = getattr(out.grad_fn, "lse_for_shift", None) if out.grad_fn is not None else None
    if lse is not None:
        tag_softmax_lse(out, lse, graph.rowptr)
    return out


class _WeightedAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, graph, ew_pair, mean):
        x = _dev_act(x, "x")
        N, d = x.shape
        out = torch.empty_like(x)
        msg = MSG_WEIGHTED if ew_pair is not None else MSG_IDENTITY
        aggr_id = AGGR_MEAN if mean else AGGR_SUM
        ew = ew_pair[0] if ew_pair is not None else None
        hub, hub_keep = graph.hub_arg("dst", d)
        rc = _lib.lib.mlgnn_csr_aggregate_fwd(
            x.data_ptr(), graph.rowptr.data_ptr(), graph.col.data_ptr(), _lib.ptr(ew), None, None, None, None,
            out.data_ptr(), None, None, None, None, N, d, _DTYPE_IDS[x.dtype], msg, EDGE_NONE, 0, aggr_id, 1.0, 1.0, None, None,
            0.0, 0, hub, _stream())
        _lib.check(rc, "mlgnn_csr_aggregate_fwd")
        ctx.graph, ctx.ew_pair, ctx.cfg = graph, ew_pair, (msg, aggr_id, N, d)
        return out

    @staticmethod
    def backward(ctx, go):
        g = ctx.graph
        msg, aggr_id, N, d = ctx.cfg
        go = _dev_act(go, "grad_out")
        gx = torch.empty_like(go)
        ew_t = ctx.ew_pair[1] if ctx.ew_pair is not None else None
        hub, hub_keep = g.hub_arg("src", d)
        rc = _lib.lib.mlgnn_csr_aggregate_bwd(
            go.data_ptr(), None, None, None, None, g.rowptr_t.data_ptr(), g.col_t.data_ptr(), g.pos_t.data_ptr(),
            g.rowptr.data_ptr(), _lib.ptr(ew_t), None, None, None, None, None, gx.data_ptr(), None, None, None, 0,
            N, d, _DTYPE_IDS[go.dtype], msg, EDGE_NONE, 0, aggr_id, 0, 1.0, 1.0, None, None, 0.0, 0, 0, hub, None, None,
            _stream())
        _lib.check(rc, "mlgnn_csr_aggregate_bwd")
        return gx, None, None, None


def weighted_mean_aggregate(x, graph, weight=None, mean=True):
    """``mean_{e: dst(e)=i} w_e x_src(e)`` (count = number of edges, not sum of weights): the SAGE
    neighbourhood reduction applied BEFORE ``lin_r`` (torch_vertex.py:279-286 by linearity).
    ``weight``: ``[E]`` / ``[E,1]`` in COO order or ``None``.  No gradient flows to ``weight``."""
    ew_pair = graph.edge_scalar(weight) if weight is not None else None
    return _WeightedAggregate.apply(x, graph, ew_pair, bool(mean))
