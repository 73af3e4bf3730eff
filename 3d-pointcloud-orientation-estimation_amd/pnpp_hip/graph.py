"""Whole-step hipGraph capture (torch.cuda.CUDAGraph on ROCm = hipGraph).

A training step of this path is ~100 short launches; once the kernels are fast the host cannot issue them as
quickly as the GPU retires them.  Everything the C ABI enqueues is stream-ordered and allocation-free, so
zero_grad + forward + loss + backward can be captured once and replayed: inputs are copied into static buffers,
centre sampling reads its counter from device memory, dropout draws from a device-side counter.  In a single
process the fused Adam launch is captured as well (step count in device memory, FlatAdam.step_dev); with data
parallelism the gradient all-reduce and the update stay outside the graph (they depend on the process group).
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch


class GraphedStep:
    def __init__(self, opt, loss_fn: Callable[..., torch.Tensor], example_inputs: Sequence[torch.Tensor], warmup: int = 3,
                 adopt_inputs: bool = False, fused_optimizer: bool = False, grad_scale: float = 1.0,
                 zero_grad_in_graph: bool = True, captured_all_reduce: Callable = None):
        """loss_fn(*inputs) -> scalar loss; `opt` is a pnpp_hip.optim.FlatAdam (its flat gradient buffer is static).

        fused_optimizer=True (single process: nothing sits between backward and the update) captures the Adam launch
        too -- opt.step_dev keeps its step count on the device -- and lets that launch clear the gradients it has
        consumed, so a replay is forward + backward + update with no memset and no eager launch; the caller then does
        NOT call opt.step().  (Measured on MI355X: back-to-back graph launches cost more than a graph followed by
        one eager launch -- the next graph's start-up hides behind the eager kernel -- so bench.py keeps Adam eager.)

        zero_grad_in_graph=False leaves the memset out of the graph: the caller's update clears the gradients it
        consumed (opt.step_dev(zero_grad=True)) instead.

        adopt_inputs=True makes `example_inputs` themselves the static input buffers (`self.static_in`): a loader that
        writes the next batch into them (H2D copy target) and then calls the step with the same tensors pays no
        device-to-device copy; any other tensor passed later is copied in as usual.

        captured_all_reduce(tensor) -> work handle: the all-reduce of the whole flat gradient is captured behind the
        backward pass (data parallelism with a capturable backend, i.e. RCCL), so the replay carries the collective and --
        with fused_optimizer -- the update: one graph launch per step."""
        self.opt = opt
        self._car = captured_all_reduce
        self.fused_optimizer = fused_optimizer
        self.static_in = list(example_inputs) if adopt_inputs else [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                      # settle workspaces / autotuned allocations before capture
                opt.zero_grad()
                self._run(loss_fn)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: with torch.distributed initialised, the process group's watchdog thread polls its events with
        # HIP calls of its own; under the default (global) mode such a call during the capture would invalidate it
        if fused_optimizer or not zero_grad_in_graph:
            opt.zero_grad()                               # the update leaves the buffer cleared from here on
        if fused_optimizer:
            opt.seed_dev_steps()
            steps_before = opt.step_count
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            if not fused_optimizer and zero_grad_in_graph:
                opt.zero_grad()
            self.static_loss = self._run(loss_fn)
            if fused_optimizer:
                opt.step_dev(grad_scale=grad_scale, zero_grad=True)
        if fused_optimizer:                               # capturing is not stepping
            opt.step_count = opt._dev_steps = steps_before

    def _run(self, loss_fn) -> torch.Tensor:
        """loss_fn may return a loss to differentiate, or (no grad_fn) one whose backward pass it has already run."""
        loss = loss_fn(*self.static_in)
        if loss.requires_grad:
            loss.backward()
        if self._car is not None:
            h = self._car(self.opt.flat_g)
            if h is not None:
                h.wait()
        return loss

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        for s, t in zip(self.static_in, inputs):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        self.graph.replay()
        if self.fused_optimizer:
            self.opt.step_count += 1
            self.opt._dev_steps = self.opt.step_count
        return self.static_loss


class GraphedSplitStep:
    """The captured step cut in two, for data parallelism: the gradient all-reduce of the layers that finish first
    overlaps the backward pass of the rest.

        graph 1: zero_grad, stage1 forward, stage2 forward, loss, backward of stage2 (down to stage1's outputs)
                 -> gradients of the parameters used by stage2 are final: their all-reduce starts (async, RCCL stream)
        graph 2: backward of stage1                                  (runs while that all-reduce is in flight)
                 -> all-reduce of the remaining (small) slice, wait for both, optimiser step (caller)

    stage1(*inputs) -> tuple of tensors; stage2(*stage1_outputs) -> scalar loss.  stage2's parameters must be the tail
    of the optimiser's flat buffer from `tail_offset` on (nn.Module.parameters() order with stage1's modules first).
    For PointNet++: stage1 = sa1 + sa2 (6 % of the parameters), stage2 = sa3 + head + loss (94 %)."""

    def __init__(self, opt, stage1: Callable, stage2: Callable, example_inputs: Sequence[torch.Tensor], tail_offset: int,
                 warmup: int = 3, adopt_inputs: bool = False, captured_all_reduce: Callable = None):
        """captured_all_reduce(tensor) -> work handle: when given, the two all-reduces are CAPTURED with the kernels in ONE
        graph (RCCL's launches become graph nodes on its own stream, forked after the backward pass of stage2 and joined
        before the end of the graph), so a replay has no host launch and no graph boundary between backward and the
        collective; `__call__` then ignores its `all_reduce` argument.  Needs a capturable backend (RCCL; gloo moves
        device tensors through the host and cannot be captured)."""
        self.opt, self.tail_offset = opt, int(tail_offset)
        self.static_in = list(example_inputs) if adopt_inputs else [t.clone() for t in example_inputs]
        self.captured_collective = captured_all_reduce is not None

        def fwd_bwd2():
            opt.zero_grad()
            mids = stage1(*self.static_in)
            mids = mids if isinstance(mids, (tuple, list)) else (mids,)
            cut = [m.detach().requires_grad_(True) if m.requires_grad else m for m in mids]   # autograd stops here
            loss = stage2(*cut)
            if loss.requires_grad:   # else stage2 has run its own backward pass (fused loss + seed)
                loss.backward()
            return mids, cut, loss

        def bwd1(mids, cut):
            pairs = [(m, c.grad) for m, c in zip(mids, cut) if m.requires_grad and c.grad is not None]
            torch.autograd.backward([m for m, _ in pairs], [g for _, g in pairs])

        def whole(all_reduce):
            g = opt.flat_g
            mids, cut, loss = fwd_bwd2()
            h1 = all_reduce(g[self.tail_offset:])
            bwd1(mids, cut)
            h2 = all_reduce(g[:self.tail_offset]) if self.tail_offset > 0 else None
            for h in (h1, h2):
                if h is not None:
                    h.wait()
            return mids, cut, loss

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                if self.captured_collective:      # the communicator's lazy set-up must not happen inside the capture
                    whole(captured_all_reduce)
                else:
                    mids, cut, _ = fwd_bwd2()
                    bwd1(mids, cut)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph1, self.graph2 = torch.cuda.CUDAGraph(), None
        if self.captured_collective:
            with torch.cuda.graph(self.graph1, capture_error_mode="thread_local"):
                mids, cut, self.static_loss = whole(captured_all_reduce)
        else:
            self.graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph1, capture_error_mode="thread_local"):
                mids, cut, self.static_loss = fwd_bwd2()
            with torch.cuda.graph(self.graph2, pool=self.graph1.pool(), capture_error_mode="thread_local"):
                bwd1(mids, cut)
        self._keep = (mids, cut)   # the autograd graph of stage1 belongs to the captured memory

    def __call__(self, *inputs: torch.Tensor, all_reduce: Callable = None) -> torch.Tensor:
        """all_reduce(tensor) -> work handle with .wait() (or None): called on the two slices of the flat gradient."""
        for s, t in zip(self.static_in, inputs):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        g = self.opt.flat_g
        self.graph1.replay()
        if self.captured_collective:
            return self.static_loss
        h1 = all_reduce(g[self.tail_offset:]) if all_reduce is not None else None
        self.graph2.replay()
        h2 = all_reduce(g[:self.tail_offset]) if all_reduce is not None and self.tail_offset > 0 else None
        for h in (h1, h2):
            if h is not None:
                h.wait()
        return self.static_loss
