"""Whole-step hipGraph capture (torch.cuda.CUDAGraph on ROCm = hipGraph).

A training step of this path is ~100 short launches; once the kernels are fast the host cannot issue them as
quickly as the GPU retires them.  Everything the C ABI enqueues is stream-ordered and allocation-free, so
zero_grad + forward + loss + backward can be captured once and replayed: inputs are copied into static buffers,
centre sampling reads its counter from device memory, dropout uses torch's graph-safe Philox offsets.  The
gradient all-reduce and the fused Adam launch stay outside the graph (they depend on host-side step counts and
on the process group).
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch


class GraphedStep:
    def __init__(self, opt, loss_fn: Callable[..., torch.Tensor], example_inputs: Sequence[torch.Tensor], warmup: int = 3,
                 adopt_inputs: bool = False):
        """loss_fn(*inputs) -> scalar loss; `opt` is a pnpp_hip.optim.FlatAdam (its flat gradient buffer is static).

        adopt_inputs=True makes `example_inputs` themselves the static input buffers (`self.static_in`): a loader that
        writes the next batch into them (H2D copy target) and then calls the step with the same tensors pays no
        device-to-device copy; any other tensor passed later is copied in as usual."""
        self.opt = opt
        self.static_in = list(example_inputs) if adopt_inputs else [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                      # settle workspaces / autotuned allocations before capture
                opt.zero_grad()
                loss_fn(*self.static_in).backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: with torch.distributed initialised, the process group's watchdog thread polls its events with
        # HIP calls of its own; under the default (global) mode such a call during the capture would invalidate it
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            opt.zero_grad()
            self.static_loss = loss_fn(*self.static_in)
            self.static_loss.backward()

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        for s, t in zip(self.static_in, inputs):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        self.graph.replay()
        return self.static_loss
