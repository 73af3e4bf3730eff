"""Epoch loop shared by the drop-in training scripts (train_single_peak_vonMises_KL.py:73-107,
train_multi_peaks_vonMises_KL.py:194-318, train_8dir_KL.py:78-152).

One step = forward -> loss -> backward -> [flat gradient all-reduce] -> [clip, on the device] -> fused Adam (which also
clears the gradients it consumed: the next iteration's zero_grad()).  When the batch shape is static and nothing in the
step needs the host (centre sampler 'device' or 'fps'), forward + loss + backward are captured once into a hipGraph and
replayed (pnpp_hip.graph.GraphedStep) -- the launch-bound eager path is only taken for a ragged last batch or a
host-side sampler.  The per-step `loss.item()` of the reference (a device sync every step) is replaced by per-sample
losses kept on the device and read once per phase; with them come the reference's per-label curves
(train_8dir_KL.py:103-110, train_multi_peaks_vonMises_KL.py:239-243).  With WORLD_SIZE > 1 every rank trains on its shard.
"""
from __future__ import annotations

import copy
import os
import time
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import dist as pdist
from . import optim


class SyntheticLoader:
    """In-memory stand-in for a DataLoader (the reference ships no dataset): yields device-resident batches, a label
    column (all zeros unless given) last, like the reference's dataset tuples."""

    def __init__(self, tensors, batch, shuffle, device, generator=None, labels=None):
        n = tensors[0].shape[0]
        labels = torch.zeros(n, dtype=torch.int64) if labels is None else labels.to(torch.int64)
        self.tensors = [t.to(device) for t in list(tensors) + [labels]]
        self.batch, self.shuffle, self.gen = batch, shuffle, generator
        self.n = n

    def __len__(self):
        return (self.n + self.batch - 1) // self.batch

    def __iter__(self):
        order = torch.randperm(self.n, generator=self.gen) if self.shuffle else torch.arange(self.n)
        order = order.to(self.tensors[0].device)
        for i in range(0, self.n, self.batch):
            sel = order[i:i + self.batch]
            yield tuple(t[sel] for t in self.tensors)


LossFn = Union[Callable, Tuple[Callable, Callable]]


def _split(loss_fn: LossFn):
    """loss_fn(model, batch) -> (B,) losses, or the pair (forward(model, batch) -> outputs, criterion(outputs, batch) ->
    (B,) losses) when the caller wants the reference's separate forward / loss timing buckets."""
    if isinstance(loss_fn, tuple):
        fwd, crit = loss_fn
        return fwd, crit, (lambda model, batch: crit(fwd(model, batch), batch))
    return None, None, loss_fn


class PhaseTimer:
    """The reference's four wall-clock buckets per batch (train_multi_peaks_vonMises_KL.py:207-237,248-252): data -> device,
    forward, loss, backward + step.  Its host clocks bracket asynchronous launches; here every bucket edge synchronises the
    device so the numbers mean what their names say -- which costs throughput, so it is opt-in (timing=True /
    PNPP_TIMING=1) and forces the eager path (phases of a replayed graph cannot be told apart)."""

    NAMES = ("data", "fwd", "loss", "bwd")

    def __init__(self, on: bool):
        self.on = on
        self.sums = {k: 0.0 for k in self.NAMES}
        self.n = 0
        self._t = 0.0

    def start(self):
        if self.on:
            torch.cuda.synchronize()
            self._t = time.time()

    def lap(self, name):
        if self.on:
            torch.cuda.synchronize()
            now = time.time()
            self.sums[name] += now - self._t
            self._t = now

    def done(self):
        self.n += 1

    def detail(self, train: bool) -> str:
        if not self.on or self.n == 0:
            return ""
        a = {k: v / self.n for k, v in self.sums.items()}
        s = f"(avg/batch: data={a['data'] * 1e3:.2f}ms fwd={a['fwd'] * 1e3:.2f}ms loss={a['loss'] * 1e3:.2f}ms"
        return s + (f" bwd={a['bwd'] * 1e3:.2f}ms)" if train else ")")


class _TrainStep:
    """forward + loss + backward (+ all-reduce, clip) + Adam for one batch; graph replay when it can, eager otherwise."""

    def __init__(self, model, loss_fn: LossFn, opt: optim.FlatAdam, clip_norm: Optional[float], world: int, use_graph: bool,
                 timer: PhaseTimer):
        self.model, self.opt, self.clip_norm, self.world, self.timer = model, opt, clip_norm, world, timer
        self.fwd, self.crit, self.loss_fn = _split(loss_fn)
        self.use_graph = use_graph and not timer.on and self._graphable(model)
        self.graphed, self.key, self.last_key = None, None, None
        self._seeds: Dict[int, torch.Tensor] = {}
        self.graph_steps = self.eager_steps = 0
        opt.zero_grad()                                   # from here on the update clears what it consumed

    @staticmethod
    def _graphable(model) -> bool:
        """A captured step cannot call the host generator: every set-abstraction level must draw its centres on the device."""
        for m in model.modules():
            if getattr(m, "group_all", True) is False and getattr(m, "sampler", "device") == "randperm":
                return False
        return True

    def _seed(self, n: int, device) -> torch.Tensor:
        """d mean(lv) / d lv = 1/n, the tensor loss.backward() would build (train_single_peak_vonMises_KL.py:83-84)."""
        s = self._seeds.get(n)
        if s is None:
            s = torch.full((n,), 1.0 / n, device=device, dtype=torch.float32)
            self._seeds[n] = s
        return s

    def _body(self, *batch) -> torch.Tensor:
        lv = self.loss_fn(self.model, batch)
        torch.autograd.backward([lv], [self._seed(lv.shape[0], lv.device)])
        return lv.detach()

    def _capture(self, batch, device):
        from .graph import GraphedStep
        from . import sampling
        static = [b.to(device).clone() if torch.is_tensor(b) else b for b in batch]
        keep = [b.detach().clone() for b in self.model.buffers()]       # warm-up passes must not count as training steps:
        streams, rng = sampling.snapshot(), torch.cuda.get_rng_state(device)   # running statistics, random streams are put back
        try:
            self.graphed = GraphedStep(self.opt, self._body, static, adopt_inputs=True, zero_grad_in_graph=False)
        except Exception as e:   # capture is an optimisation, never a requirement
            print(f"[trainer] hipGraph capture failed, staying eager: {type(e).__name__}: {e}")
            self.graphed, self.use_graph = None, False
            torch.cuda.synchronize()
        with torch.no_grad():
            for b, k in zip(self.model.buffers(), keep):
                b.copy_(k)
        sampling.restore(streams)
        torch.cuda.set_rng_state(rng, device)
        self.opt.zero_grad()

    def __call__(self, batch: Sequence, device) -> torch.Tensor:
        key = tuple((tuple(b.shape), b.dtype) if torch.is_tensor(b) else None for b in batch)
        if self.use_graph and self.graphed is None and key == self.last_key:   # the second batch of this shape: it is static
            self.key = key
            self._capture(batch, device)
        self.last_key = key
        t = self.timer
        if self.graphed is not None and key == self.key:
            lv = self.graphed(*batch).clone()             # H2D (or D2D) straight into the graph's static inputs, then replay
            self.graph_steps += 1
        else:
            t.start()
            batch = tuple(b.to(device, non_blocking=True) if torch.is_tensor(b) else b for b in batch)
            t.lap("data")
            if self.fwd is not None and t.on:
                out = self.fwd(self.model, batch)
                t.lap("fwd")
                lv = self.crit(out, batch)
                t.lap("loss")
                torch.autograd.backward([lv], [self._seed(lv.shape[0], lv.device)])
                lv = lv.detach()
            else:
                lv = self._body(*batch)
            self.eager_steps += 1
        pdist.all_reduce_flat_grad(self.opt.flat_g)
        if self.clip_norm is not None:                    # norm of the MEAN gradient, on the device (no .item())
            self.opt.clip_grad_norm_(self.clip_norm)
        self.opt.step(grad_scale=1.0 / self.world, zero_grad=True)
        t.lap("bwd")
        t.done()
        return lv


class _PhaseLog:
    """Per-sample losses (device) and labels of one phase; reduced on the host once, at the end of the phase."""

    def __init__(self, n_labels: int):
        self.n_labels = n_labels
        self.lv: List[torch.Tensor] = []
        self.lab: List[torch.Tensor] = []

    def add(self, lv: torch.Tensor, labels: Optional[torch.Tensor]):
        self.lv.append(lv)
        if labels is not None:
            self.lab.append(labels.detach().reshape(-1))

    def reduce(self, device):
        """-> (mean over every rank's samples, per-label means [nan where a label has no sample])."""
        nl = self.n_labels
        pack = np.zeros(2 + 2 * nl)
        if self.lv:
            v = torch.cat(self.lv).double().cpu().numpy()                 # the one device read of the phase
            pack[0], pack[1] = v.sum(), v.size
            if nl and self.lab:
                lab = torch.cat([l.cpu() for l in self.lab]).numpy().astype(np.int64)
                ok = (lab >= 0) & (lab < nl)
                pack[2:2 + nl] = np.bincount(lab[ok], weights=v[ok], minlength=nl)
                pack[2 + nl:] = np.bincount(lab[ok], minlength=nl)
        if pdist.world_size() > 1:   # one tiny all-reduce per epoch and phase: all ranks agree on the history
            import torch.distributed as tdist
            t = torch.from_numpy(pack).to(device)
            tdist.all_reduce(t)
            pack = t.cpu().numpy()
        total = float(pack[0] / max(pack[1], 1.0))
        with np.errstate(invalid="ignore", divide="ignore"):
            per = np.where(pack[2 + nl:] > 0, pack[2:2 + nl] / pack[2 + nl:], np.nan)
        return total, [float(x) for x in per]


def fit(model: torch.nn.Module, loss_fn: LossFn, loaders: Dict[str, Iterable], epochs: int, lr: float,
        device: torch.device, clip_norm: Optional[float] = None, log: Callable = print, label: str = "KL",
        label_index: Optional[int] = None, n_labels: int = 0, use_graph: Optional[bool] = None, timing: Optional[bool] = None):
    """Trains with FlatAdam; returns (history, best_state, best_val_epoch).

    loss_fn(model, batch) -> per-sample loss vector (B,) on the device (or a (forward, criterion) pair, see _split);
    batch tensors are already on `device` when it is called.  label_index: position of the class-index column in the
    batch tuple (the reference's datasets yield it last); with n_labels > 0 the history also carries
    hist["labels"][i][phase] -- the per-label epoch means the reference scripts plot and summarise.
    best_state is a deep copy taken at the best validation epoch (the reference keeps live references, so its
    "best" checkpoint is in fact the last one -- train_single_peak_vonMises_KL.py:90; the copy is deliberate).
    """
    world = pdist.world_size()
    if world > 1 and os.environ.get("PNPP_SYNC_BN") == "1":
        # SyncBN (off by default): BatchNorm statistics pooled over the ranks, so the global batch normalises like the reference's
        # single process on the concatenated batch (22 small all-reduces per step; DDP-style per-rank statistics otherwise)
        pdist.enable_sync_batchnorm()
        if pdist.dist.get_backend() != "nccl" and use_graph is None:
            use_graph = False                     # a host-staged exchange cannot be captured into a hipGraph
    if use_graph is None:
        use_graph = os.environ.get("PNPP_NO_GRAPH") != "1"
    if timing is None:
        timing = os.environ.get("PNPP_TIMING") == "1"
    opt = optim.FlatAdam(model.parameters(), lr=lr)
    pdist.broadcast_flat(opt.flat_p)
    _, _, plain_loss = _split(loss_fn)
    hist = {"train": [], "val": [], "labels": [{"train": [], "val": []} for _ in range(n_labels)], "timing": [],
            "seconds": {"train": [], "val": []}, "samples": {"train": [], "val": []}}
    best_val, best_state, best_ep = float("inf"), None, None
    t0 = time.time()
    train_timer = PhaseTimer(timing)
    step = _TrainStep(model, loss_fn, opt, clip_norm, world, use_graph, train_timer)
    for ep in range(1, epochs + 1):
        ep_t = time.time()
        details = {}
        for phase in ("train", "val"):
            if phase not in loaders:
                continue
            model.train() if phase == "train" else model.eval()
            plog = _PhaseLog(n_labels)
            timer = train_timer if phase == "train" else PhaseTimer(timing)
            if phase == "train":
                timer.sums, timer.n = {k: 0.0 for k in PhaseTimer.NAMES}, 0
            limit = _common_steps(loaders[phase], device) if phase == "train" else None
            ph_t = time.time()
            for step_i, batch in enumerate(loaders[phase]):
                if limit is not None and step_i >= limit:
                    break   # every rank takes the same number of optimiser steps (one all-reduce each)
                labels = batch[label_index] if label_index is not None else None
                if phase == "train":
                    lv = step(batch, device)
                else:
                    lv = _eval_batch(model, loss_fn, batch, device, timer)
                plog.add(lv, labels)
            avg, per = plog.reduce(device)                   # reads the losses back: the phase's work is finished here
            hist["seconds"][phase].append(time.time() - ph_t)
            hist["samples"][phase].append(sum(int(t.shape[0]) for t in plog.lv))
            hist[phase].append(avg)
            for i, v in enumerate(per):
                hist["labels"][i][phase].append(v)
            details[phase] = timer.detail(phase == "train")
            if phase == "val" and avg < best_val:
                best_val, best_ep = avg, ep
                best_state = copy.deepcopy(model.state_dict())
        va = hist["val"][-1] if hist["val"] else float("nan")
        hist["timing"].append(details)
        if pdist.rank() == 0:
            el = time.time() - ep_t
            eta = (time.time() - t0) / ep * (epochs - ep)
            extra = "".join(f" | {ph}: {d}" for ph, d in details.items() if d)
            log(f"Ep {ep:03}/{epochs} Train {hist['train'][-1]:.4f} Val {va:.4f} | {label} | Time: {el:.1f}s | "
                f"ETA: {eta / 60:.1f}m{extra}")
    hist["steps"] = {"graph": step.graph_steps, "eager": step.eager_steps}
    if best_state is None:
        best_state = copy.deepcopy(model.state_dict())
    return hist, best_state, best_ep


def _eval_batch(model, loss_fn: LossFn, batch, device, timer: Optional[PhaseTimer] = None) -> torch.Tensor:
    fwd, crit, plain = _split(loss_fn)
    t = timer if timer is not None else PhaseTimer(False)
    t.start()
    batch = tuple(b.to(device, non_blocking=True) if torch.is_tensor(b) else b for b in batch)
    t.lap("data")
    with torch.no_grad():
        if fwd is not None and t.on:
            out = fwd(model, batch)
            t.lap("fwd")
            lv = crit(out, batch)
            t.lap("loss")
        else:
            lv = plain(model, batch)
    t.done()
    return lv.detach()


def evaluate(model, loss_fn: LossFn, loader, device, label_index: Optional[int] = None, n_labels: int = 0):
    """Mean per-sample loss over `loader` (every rank's samples); with n_labels > 0 returns (mean, per-label means)."""
    model.eval()
    plog = _PhaseLog(n_labels)
    for batch in loader:
        plog.add(_eval_batch(model, loss_fn, batch, device), batch[label_index] if label_index is not None else None)
    total, per = plog.reduce(device)
    return (total, per) if n_labels else total


def _common_steps(loader, device):
    """With more than one rank: the smallest number of batches any rank's shard yields (shards differ by at most one
    sample, so at most one batch); None for a single process."""
    if pdist.world_size() <= 1:
        return None
    import torch.distributed as tdist
    n = torch.tensor([len(loader)], dtype=torch.int64, device=device)
    tdist.all_reduce(n, op=tdist.ReduceOp.MIN)
    return int(n.item())


def env_path(name: str, default: str):
    from pathlib import Path
    return Path(os.environ.get(name, default))
