"""Step glue on ONE flat parameter / gradient buffer.

The reference's loops call opt.zero_grad(), loss.backward(), [clip_grad_norm_], opt.step() over ~60 small
tensors (train_single_peak_vonMises_KL.py:80-85, train_multi_peaks_vonMises_KL.py:221-236).  Here every
parameter is a view into one contiguous buffer and every .grad a view into another, so zero_grad is one
memset, the optimiser one kernel launch, and data parallelism one all-reduce.
"""
from __future__ import annotations

import math
from typing import Iterable, Optional

import torch

from . import _lib as L
from .ops import _stream


class FlatAdam:
    """torch.optim.Adam(params, lr, betas, eps) semantics (no amsgrad / weight decay), one fused launch."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, fused_grads=True):
        """fused_grads: register every parameter's slice of the flat gradient buffer as the destination the
        backward kernels write to directly; autograd then has no per-parameter accumulation kernels to launch.  The
        first use of a parameter after zero_grad()/step() overwrites its slice; any further use in the same window
        (micro-batch accumulation, two passes summed into one loss, shared weights) is detected when its forward pass
        asks for the destination (ops.grad_sink) and goes through autograd's accumulation into the same memory."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam runs on the GPU only (no CPU fallback exists in this package)")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        total = sum(p.numel() for p in self.params)
        self.flat_p = torch.empty(total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self._views = []
        off = 0
        for p in self.params:
            n = p.numel()
            self.flat_p[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + n].view(p.shape)
            g = self.flat_g[off:off + n].view(p.shape)
            p.grad = g
            if fused_grads:
                p._pnpp_grad_sink = g
            self._views.append(g)
            off += n
        self.step_count = 0
        self._step_state = torch.zeros(2, device=dev, dtype=torch.int64)   # step_dev(): [steps taken, ticket]
        self._scratch = torch.empty(1024 * 8 + 8, device=dev, dtype=torch.uint8)
        self._ss = torch.zeros(1, device=dev, dtype=torch.float64)          # sum of squares of flat_g (clip_grad_norm_)
        self._pending_clip: Optional[float] = None

    @property
    def numel(self) -> int:
        return self.flat_p.numel()

    def offset_of(self, param: torch.nn.Parameter) -> int:
        """Start of `param` in the flat buffers (parameters keep the order they were passed in)."""
        off = 0
        for p in self.params:
            if p is param:
                return off
            off += p.numel()
        raise KeyError("parameter is not managed by this optimiser")

    def zero_grad(self) -> None:
        """One memset; autograd then accumulates in place into the flat buffer."""
        self.flat_g.zero_()
        for p, g in zip(self.params, self._views):
            if p.grad is not g:
                p.grad = g
        self._release_sinks()

    def _release_sinks(self) -> None:
        """A new accumulation window: the first backward write of every parameter may overwrite again (ops.grad_sink)."""
        for p in self.params:
            p._pnpp_sink_claimed = False

    def _sumsq(self) -> torch.Tensor:
        """Sum of squares of the flat gradient buffer into a persistent device word (float64); one stream-ordered launch pair."""
        L.check(L.lib().pnpp_sumsq(self.flat_g.data_ptr(), self.flat_g.numel(), self._ss.data_ptr(), self._scratch.data_ptr(),
                                   self._scratch.numel(), _stream()))
        return self._ss

    def grad_norm(self) -> torch.Tensor:
        """L2 norm of the whole gradient as a 0-dim float64 tensor on the device (no host sync)."""
        return self._sumsq().clone().sqrt_()[0]

    def clip_grad_norm_(self, max_norm: float, return_norm: bool = True) -> Optional[torch.Tensor]:
        """torch.nn.utils.clip_grad_norm_(parameters, max_norm) semantics (train_multi_peaks_vonMises_KL.py:235) with no
        host round trip: the sum of squares is reduced on the device now and the NEXT step()/step_dev() reads it there and
        folds min(1, max_norm / (norm + 1e-6)) into its gradient scale (the gradient buffer itself is left untouched --
        the update consumes it once).  Under data parallelism call it after the all-reduce: the norm that is clipped is
        that of grad_scale * flat_g, i.e. of the mean gradient.  Returns the buffer's own L2 norm as a 0-dim float64 device
        tensor (multiply by grad_scale for the mean gradient's); reading it is the caller's sync, not this method's.
        return_norm=False skips that tensor (two small launches) when the caller does not look at the norm."""
        self._sumsq()
        self._pending_clip = float(max_norm)
        return self._ss.sqrt()[0] if return_norm else None

    def step(self, grad_scale: float = 1.0, zero_grad: bool = False) -> None:
        """zero_grad=True clears the flat gradient buffer in the same launch (the next iteration's zero_grad())."""
        self.step_count += 1
        self._release_sinks()
        clip, self._pending_clip = self._pending_clip, None
        args = (self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                self.flat_p.numel(), self.step_count, self.lr, self.betas[0], self.betas[1], self.eps, float(grad_scale))
        if clip is not None:
            L.check(L.lib().pnpp_adam_step_clip(*args, self._ss.data_ptr(), clip, int(bool(zero_grad)), _stream()))
        else:
            fn = L.lib().pnpp_adam_step_zero if zero_grad else L.lib().pnpp_adam_step
            L.check(fn(*args, _stream()))

    def seed_dev_steps(self) -> None:
        """Copies the host step count into the device word step_dev() reads, if they differ (not capturable: call it
        before a capture starts)."""
        if getattr(self, "_dev_steps", None) != self.step_count:
            self._step_state.copy_(torch.tensor([self.step_count, 0], dtype=torch.int64))
            self._dev_steps = self.step_count

    def step_dev(self, grad_scale: float = 1.0, zero_grad: bool = False) -> None:
        """step() with the step count kept in device memory, so the launch can be captured in a hipGraph and replayed
        (pnpp_hip.graph.GraphedStep(fused_optimizer=True)); `zero_grad` clears the flat gradient buffer in the same launch.
        The device count is re-seeded from `step_count` whenever the two disagree (first use, after eager step()s);
        callers that replay a captured step_dev bump `step_count` themselves."""
        self.seed_dev_steps()
        self.step_count += 1
        self._release_sinks()
        self._dev_steps = self.step_count
        clip, self._pending_clip = self._pending_clip, None
        head = (self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                self.flat_p.numel(), self._step_state.data_ptr(), self.lr, self.betas[0], self.betas[1], self.eps, float(grad_scale))
        if clip is not None:
            L.check(L.lib().pnpp_adam_step_dev_clip(*head, self._ss.data_ptr(), clip, int(bool(zero_grad)), _stream()))
        else:
            L.check(L.lib().pnpp_adam_step_dev(*head, int(bool(zero_grad)), _stream()))
