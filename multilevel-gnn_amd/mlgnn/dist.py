"""Data-parallel training over the GPUs of one node: one process per GPU, graphs sharded by
rank, ONE RCCL all-reduce per step over a single flat fp32 gradient buffer.

The reference has no distributed code at all; this is the exchange step
BASELINE.json's north_star adds.  Parameter sets are 1-37 MB (section 5): on the xGMI mesh that
is latency-dominated, so the gradients live contiguously in one bucket (``p.grad`` are views into
it -- autograd accumulates in place, no flatten/unflatten copies) and a single ``all_reduce``
moves them.  Works with the ``nccl`` (= RCCL) backend on GPUs and ``gloo`` on CPU (tests).
"""
import torch
import torch.distributed as dist


SLOT_ALIGN = 4            # elements: 16 bytes of fp32


def _collective(fn, tensor, group, **kw):
    """``fn(tensor, group=group, **kw)`` in place.  RCCL takes the device buffer as it is; the ``gloo`` rehearsal backend
    (several ranks sharing the single GPU of a test box: bench.py's MLGNN_BENCH_BACKEND=gloo, tests/test_bench_gpu.py)
    gets a host copy, so that path does not depend on gloo having been built with device support."""
    if tensor.is_cuda and dist.get_backend(group) == "gloo":
        host = tensor.cpu()
        fn(host, group=group, **kw)
        tensor.copy_(host)
    else:
        fn(tensor, group=group, **kw)


class FlatGradBucket:
    def __init__(self, module, process_group=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        # every parameter's slot starts on a 16-byte boundary (SLOT_ALIGN elements): the kernels read weights with
        # 16-byte loads straight out of the flat parameter buffer that mirrors this layout (mlgnn.optim.FlatAdam);
        # learnable_pca_params [25015, 2] alone would misalign everything behind it.  The gaps stay zero.
        self.offsets = [0]
        for p in self.params:
            self.offsets.append(self.offsets[-1] + (p.numel() + SLOT_ALIGN - 1) // SLOT_ALIGN * SLOT_ALIGN)
        total = self.offsets[-1]
        self._gap_zeros = torch.zeros(SLOT_ALIGN, dtype=dt, device=dev)
        # gradients, then one "reached" flag per parameter: the flags travel with the gradients in the ONE all-reduce,
        # so every rank learns the union of what any rank's backward reached (a parameter reached on rank 0 only is
        # stepped, with the averaged gradient, on every rank -- or the replicas would drift apart silently)
        self.flat_all = torch.zeros(total + len(self.params), dtype=dt, device=dev)
        self.flat = self.flat_all[:total]
        self.live = self.flat_all[total:]
        self.live.fill_(1)
        self.group = process_group
        self.views = []
        self.reached = [True] * len(self.params)       # which parameters the last collected LOCAL backward reached
        self._live_local = (tuple(self.reached), self.live.clone())
        for p, off in zip(self.params, self.offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            self.views.append(p.grad)
            # a backward that produces this parameter's whole gradient in one kernel may write it here directly and hand
            # autograd an alias (mlgnn.dense._SkinnyLinear: the 172 MB head weight at config/kirc.yaml) -- collect() then
            # finds the gradient already in place
            p._mlgnn_grad_slot = self.flat[off:off + p.numel()]

    # Two ways to fill the bucket.  (a) zero() before backward: autograd accumulates into the views in place -- one
    # small add kernel per parameter.  (b) release() before backward, collect() after it: autograd hands over fresh
    # gradient tensors and ONE multi-tensor copy moves them into the bucket (60 launches fewer per step for the
    # benchmark model); the views are the parameters' .grad again afterwards, for the optimizer.
    def release(self):
        for p in self.params:
            p.grad = None

    def collect(self):
        # runs of consecutive parameters with a fresh gradient tensor are concatenated straight into their slice of
        # the flat buffer: ONE launch per run (`torch.cat(..., out=)`, public API) -- one launch in all when every
        # parameter was reached, as in every shipped model
        # (the alignment gap behind a parameter is written with zeros by the same launch)
        run, run_start = [], 0

        def flush(end):
            if run:
                torch.cat(run, out=self.flat[run_start:end])
                del run[:]

        for i, (p, v) in enumerate(zip(self.params, self.views)):
            off, nxt = self.offsets[i], self.offsets[i + 1]
            self.reached[i] = p.grad is not None
            fresh = p.grad is not None and p.grad.data_ptr() != v.data_ptr()
            if fresh and p.grad.dtype == self.flat.dtype:
                if not run:
                    run_start = off
                run.append(p.grad.reshape(-1))
                if nxt - off > p.numel():
                    run.append(self._gap_zeros[:nxt - off - p.numel()])
            else:
                flush(off)
                if p.grad is None:
                    v.zero_()                      # parameter not reached by this backward: zero in the flat buffer
                elif fresh:
                    v.copy_(p.grad)                # (a gradient of another dtype: converting copy)
        flush(self.offsets[-1])
        for p, v in zip(self.params, self.views):
            p.grad = v
        key = tuple(self.reached)
        if key != self._live_local[0]:
            self._live_local = (key, torch.tensor([1.0 if r else 0.0 for r in key], dtype=self.live.dtype,
                                                  device=self.live.device))
        self.live.copy_(self._live_local[1])       # (re-written every step: the all-reduce below leaves sums in it)
        # (torch.optim optimizers then see a zero gradient for an unreached parameter where the reference's
        # zero_grad(set_to_none) would make them skip it: with weight decay that parameter decays.
        # mlgnn.optim.FlatAdam reads `reached` and skips it like torch does.)

    def zero(self):
        """Use instead of ``optimizer.zero_grad()`` (which would drop the views)."""
        self.flat.zero_()

    def all_reduce_mean(self):
        """Sum over ranks, divide by world size: mean of per-rank mean losses = global mean for
        equal per-rank batch sizes."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(self.group)
        if world == 1:
            return
        _collective(dist.all_reduce, self.flat_all, self.group, op=dist.ReduceOp.SUM)
        self.flat_all.div_(world)                  # flags: (ranks that reached the parameter) / world, > 0 = live

    def reached_anywhere(self):
        """Per-parameter bools after :meth:`all_reduce_mean`: reached by some rank's backward (reads the device)."""
        return [v > 0 for v in self.live.tolist()]

    def check_views(self):
        """True while every ``p.grad`` still aliases the bucket."""
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
                return False
        return True


def broadcast_parameters(module, src=0, process_group=None):
    """Rank ``src``'s initial weights everywhere (one flat broadcast)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers() if b.dtype.is_floating_point]
    flat = torch.cat([t.reshape(-1) for t in tensors])
    _collective(dist.broadcast, flat, process_group, src=src)
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
