"""Optimizer step of the reference's training loop on one flat buffer (``mlgnn_adam_step``).

Reference: ``train.py:112-114`` -- ``torch.optim.Adam(model.parameters(), lr, betas, weight_decay=wd)`` and
``StepLR(step_size, gamma)`` -- and ``:63-66`` -- ``loss.backward(); clip_grad_norm_(..., 20); optimizer.step()``.
:class:`FlatAdam` lays parameters, gradients and both moments out contiguously (the gradient side is
:class:`mlgnn.dist.FlatGradBucket`, so the data-parallel all-reduce and the optimizer share one buffer) and runs
clipping + Adam for the whole model as two kernel launches; :class:`StepLR` is the host-side schedule.
"""
import math

import torch

from . import _lib
from .dist import FlatGradBucket

class FlatAdam:
    """``torch.optim.Adam`` (L2 ``weight_decay``, no amsgrad) + optional ``clip_grad_norm_(max_norm)`` over a module
    whose parameters are re-laid into one flat fp32 buffer (``p.data`` become views of it; ``state_dict`` is
    unaffected).  Usage per step::

        bucket.release(); loss.backward(); bucket.collect(); bucket.all_reduce_mean(); opt.step()

    with ``bucket = opt.bucket``.  Parameters the backward did not reach are skipped like ``grad is None`` in torch;
    the kernel reads "reached" from the device flags behind the gradients in the bucket, which the all-reduce has turned
    into "reached on any rank" -- all ranks step the same parameters (a parameter only some ranks reach gets the
    averaged gradient everywhere, like DDP with ``find_unused_parameters``).  Any number of parameters, any
    interleaving of reached and unreached ones.
    One global step count: a parameter that is reached only in some steps uses it too (torch counts per parameter);
    the shipped models reach a parameter either always or never."""

    def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_grad_norm=None,
                 process_group=None):
        self.bucket = FlatGradBucket(module, process_group)
        params = self.bucket.params
        dev = params[0].device
        if not all(p.dtype == torch.float32 and p.device == dev for p in params):
            raise TypeError("FlatAdam wants fp32 parameters on one device")
        # the bucket's slot layout (every parameter on a 16-byte boundary, zero gaps): parameters, gradients and
        # moments line up element for element
        offs = self.bucket.offsets
        self.sizes = [b - a for a, b in zip(offs[:-1], offs[1:])]
        self.flat_p = torch.zeros(offs[-1], dtype=torch.float32, device=dev)
        for p, off in zip(params, offs):
            view = self.flat_p[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.param_groups = [dict(lr=lr, initial_lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.clip = float(clip_grad_norm) if clip_grad_norm else 0.0
        self.step_count = 0
        self._ws = torch.zeros(int(_lib.lib.mlgnn_adam_workspace_floats()), dtype=torch.float32, device=dev) \
            if dev.type == "cuda" else None
        self._offsets = torch.tensor(offs, dtype=torch.int64, device=dev)

    @property
    def grad_norm(self):
        """Total gradient norm of the last clipped step (device scalar; what ``clip_grad_norm_`` returns)."""
        return self._ws[256]

    def zero_grad(self, set_to_none=True):
        self.bucket.release()

    def step(self):
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        self.step_count += 1
        t = self.step_count
        step_size = g["lr"] / (1.0 - b1 ** t)
        bias2_sqrt = math.sqrt(1.0 - b2 ** t)
        rc = _lib.lib.mlgnn_adam_step(self.flat_p.data_ptr(), self.bucket.flat.data_ptr(), self.exp_avg.data_ptr(),
                                      self.exp_avg_sq.data_ptr(), self.flat_p.numel(), self._offsets.data_ptr(),
                                      self.bucket.live.data_ptr(), len(self.sizes), self.clip, b1, b2, g["eps"],
                                      g["weight_decay"], step_size, bias2_sqrt, self._ws.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mlgnn_adam_step")
        # the kernel wrote parameters, moments and (when clipping) gradients behind torch's back: bump the version
        # counters (shared by every view of the flat buffers) so that version-keyed caches and autograd's
        # saved-tensor checks see the update
        torch.autograd.graph.increment_version(self.flat_p)
        if self.clip > 0:
            torch.autograd.graph.increment_version(self.bucket.flat)


class StepLR:
    """``torch.optim.lr_scheduler.StepLR``: ``lr = initial_lr * gamma ** (epoch // step_size)`` after ``epoch`` calls of
    :meth:`step` (``train.py:114,205``)."""

    def __init__(self, optimizer, step_size, gamma=0.1):
        self.optimizer, self.step_size, self.gamma, self.last_epoch = optimizer, int(step_size), gamma, 0

    def step(self):
        self.last_epoch += 1
        for g in self.optimizer.param_groups:
            g["lr"] = g["initial_lr"] * self.gamma ** (self.last_epoch // self.step_size)

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]
