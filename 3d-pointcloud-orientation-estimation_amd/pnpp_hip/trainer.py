"""Epoch loop shared by the drop-in training scripts (train_single_peak_vonMises_KL.py:73-107,
train_multi_peaks_vonMises_KL.py:194-318, train_8dir_KL.py:78-152).

One step = zero_grad (one memset) -> forward -> loss -> backward -> [flat gradient all-reduce] ->
[clip] -> fused Adam.  The per-step `loss.item()` of the reference (a device sync every step) is replaced by a
running sum on the device that is read once per phase.  With WORLD_SIZE > 1 every rank trains on its own shard.
"""
from __future__ import annotations

import copy
import os
import time
from typing import Callable, Dict, Iterable, Optional

import torch

from . import dist as pdist
from . import optim


class SyntheticLoader:
    """In-memory stand-in for a DataLoader (the reference ships no dataset): yields device-resident batches."""

    def __init__(self, tensors, batch, shuffle, device, generator=None):
        self.tensors = [t.to(device) for t in tensors]
        self.batch, self.shuffle, self.gen = batch, shuffle, generator
        self.n = self.tensors[0].shape[0]

    def __len__(self):
        return (self.n + self.batch - 1) // self.batch

    def __iter__(self):
        order = torch.randperm(self.n, generator=self.gen) if self.shuffle else torch.arange(self.n)
        order = order.to(self.tensors[0].device)
        for i in range(0, self.n, self.batch):
            sel = order[i:i + self.batch]
            yield tuple(t[sel] for t in self.tensors)


def fit(model: torch.nn.Module, loss_fn: Callable, loaders: Dict[str, Iterable], epochs: int, lr: float,
        device: torch.device, clip_norm: Optional[float] = None, log: Callable = print, label: str = "KL"):
    """Trains with FlatAdam; returns (history, best_state, best_val_epoch).

    loss_fn(model, batch) -> per-sample loss vector (B,) on the device; batch tensors are already on `device`.
    best_state is a deep copy taken at the best validation epoch (the reference keeps live references, so its
    "best" checkpoint is in fact the last one -- train_single_peak_vonMises_KL.py:90; the copy is deliberate).
    """
    world = pdist.world_size()
    opt = optim.FlatAdam(model.parameters(), lr=lr)
    pdist.broadcast_flat(opt.flat_p)
    hist = {"train": [], "val": []}
    best_val, best_state, best_ep = float("inf"), None, None
    t0 = time.time()
    for ep in range(1, epochs + 1):
        for phase in ("train", "val"):
            if phase not in loaders:
                continue
            model.train() if phase == "train" else model.eval()
            total = torch.zeros((), device=device, dtype=torch.float64)
            cnt = 0
            limit = _common_steps(loaders[phase], device) if phase == "train" else None
            for step_i, batch in enumerate(loaders[phase]):
                if limit is not None and step_i >= limit:
                    break   # every rank takes the same number of optimiser steps (one all-reduce each)
                batch = tuple(b.to(device, non_blocking=True) if torch.is_tensor(b) else b for b in batch)
                if phase == "train":
                    opt.zero_grad()
                    lv = loss_fn(model, batch)
                    lv.mean().backward()
                    pdist.all_reduce_flat_grad(opt.flat_g)
                    if clip_norm is not None:
                        opt.clip_grad_norm_(clip_norm * world)   # flat_g holds the SUM over ranks until step() scales it
                    opt.step(grad_scale=1.0 / world)
                else:
                    with torch.no_grad():
                        lv = loss_fn(model, batch)
                total += lv.detach().double().sum()
                cnt += lv.shape[0]
            avg = _global_mean(total, cnt, device)
            hist[phase].append(avg)
            if phase == "val" and avg < best_val:
                best_val, best_ep = avg, ep
                best_state = copy.deepcopy(model.state_dict())
        va = hist["val"][-1] if hist["val"] else float("nan")
        if pdist.rank() == 0:
            log(f"Ep {ep:03}/{epochs} Train {hist['train'][-1]:.4f} Val {va:.4f} | {label} | elapsed {time.time() - t0:.1f}s")
    if best_state is None:
        best_state = copy.deepcopy(model.state_dict())
    return hist, best_state, best_ep


def evaluate(model, loss_fn, loader, device) -> float:
    model.eval()
    total = torch.zeros((), device=device, dtype=torch.float64)
    cnt = 0
    with torch.no_grad():
        for batch in loader:
            batch = tuple(b.to(device, non_blocking=True) if torch.is_tensor(b) else b for b in batch)
            lv = loss_fn(model, batch)
            total += lv.double().sum()
            cnt += lv.shape[0]
    return _global_mean(total, cnt, device)


def _common_steps(loader, device):
    """With more than one rank: the smallest number of batches any rank's shard yields (shards differ by at most one
    sample, so at most one batch); None for a single process."""
    if pdist.world_size() <= 1:
        return None
    import torch.distributed as tdist
    n = torch.tensor([len(loader)], dtype=torch.int64, device=device)
    tdist.all_reduce(n, op=tdist.ReduceOp.MIN)
    return int(n.item())


def _global_mean(total: torch.Tensor, cnt: int, device) -> float:
    """Mean over every rank's samples (one tiny all-reduce per epoch and phase): all ranks then agree on the history,
    and with it on the best-validation epoch whose weights they keep."""
    pair = torch.stack([total.double().reshape(()), torch.tensor(float(cnt), dtype=torch.float64, device=total.device)])
    if pdist.world_size() > 1:
        import torch.distributed as tdist
        tdist.all_reduce(pair)
    return float(pair[0]) / max(float(pair[1]), 1.0)


def env_path(name: str, default: str):
    from pathlib import Path
    return Path(os.environ.get(name, default))
