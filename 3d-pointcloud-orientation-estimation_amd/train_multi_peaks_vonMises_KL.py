#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""train_multi_peaks_vonMises_KL.py -- drop-in for the reference script of the same name.

Keeps kl_von_mises(mu_p, kappa_p, mu_q, kappa_q), match_loss(mu_pred, kappa_pred, w_pred, vm_gt, _, K_gt),
plot_curve / plot_label_curve / plot_total_curve, write_summary_txt and main(); outputs RES/mvM_best.pth, RES/results.txt in
the reference's format (lines 127-146, per-category rows filled from the per-label epoch means), FIGS/loss_<category>.png,
FIGS/loss_total.png and FIGS/loss_overview.png (lines 292-299).  match_loss is one HIP launch for the whole batch: K x K clamped/wrapped KL cost, optimal
assignment and the weighted mean, value and gradients, with no per-sample host round trip (the reference crosses
to the host for scipy's linear_sum_assignment once per sample, lines 74-75).
"""
import argparse
import math
import os
import random
import re
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader

from dataloader_multi_peak_vonMises import PointCloudDatasetMvM
from models.pointnet_pp_mvM import PointNetPPMvM
from pnpp_hip import sampling, dist as pdist, ops, trainer

ROOT = trainer.env_path("PNPP_ROOT", "/home/pablo/ForwardNet/data/MN40_multi_peak_vM_gt")
PLY_ROOT = trainer.env_path("PNPP_PLY_ROOT", "/home/pablo/ForwardNet/data/full_mn40_normal_resampled_2d_rotated_ply")
RES = trainer.env_path("PNPP_RES", "/home/pablo/ForwardNet/results/multi_peak_vonMises_KL_1012_1")
FIGS = RES / "figs"

NUM_POINTS = int(os.environ.get("PNPP_NUM_POINTS", 10_000))
BATCH = int(os.environ.get("PNPP_BATCH", 16))
EPOCHS = int(os.environ.get("PNPP_EPOCHS", 100))
LR = float(os.environ.get("PNPP_LR", 1e-3))
SEED = int(os.environ.get("PNPP_SEED", 42))
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def kl_von_mises(mu_p, kappa_p, mu_q, kappa_q):
    """Elementwise multi-peak KL (reference lines 38-52): kappa clamped to [1e-6, 500], angle wrapped to [-pi, pi).
    Helper for analysis; the training path evaluates the same formula inside match_loss's kernel.  Uses the
    overflow-free form log I0(k) = k + log(i0e(k)) so kappa beyond fp32's i0 range stays finite."""
    kp = torch.clamp(kappa_p, 1e-6, 500.0)
    kq = torch.clamp(kappa_q, 1e-6, 500.0)
    a = torch.special.i1e(kp) / torch.special.i0e(kp)
    d = (mu_p - mu_q + math.pi) % (2 * math.pi) - math.pi
    log_ratio = (kq + torch.log(torch.special.i0e(kq))) - (kp + torch.log(torch.special.i0e(kp)))
    return log_ratio + a * (kp - kq * torch.cos(d))


def match_loss(mu_pred, kappa_pred, w_pred, vm_gt, _, K_gt):
    """Per-sample matched loss (reference lines 54-81); fifth argument unused, as in the reference."""
    return ops.match_loss(mu_pred, kappa_pred, w_pred, vm_gt, K_gt)


def _sanitize(name: str) -> str:
    return re.sub(r"[^\w\-]+", "_", name.strip())


# results.txt: the output contract of the reference's script (its lines 127-146) as data -- one template per line kind.
_SUMMARY_HEADER = "=== Multi-Peak von Mises KL Summary ==="
_SUMMARY_OPTIONAL = (("best_val_epoch", "Best Total Val Epoch: {}"), ("test_kl", "Test KL: {:.6f}"))
_SUMMARY_SECTION = "-- Per-Category (last epoch) --"
_SUMMARY_ROW = "[{label}] Train={train} Val={val}"


def _six_decimals(value) -> str:
    """A number with six decimals; anything that is not a number reads "nan"."""
    try:
        return format(float(value), ".6f")
    except (TypeError, ValueError):
        return "nan"


def write_summary_txt(path_txt: Path, categories, hist, test_kl=None, best_val_epoch=None):
    """results.txt, byte for byte in the layout the reference writes: header, the optional best-epoch / test lines, then
    one "[label] Train=… Val=…" row for TOTAL and for every category, taken at the last epoch TOTAL has (a category whose
    curve is empty reads nan)."""
    given = {"best_val_epoch": best_val_epoch, "test_kl": test_kl}
    lines = [_SUMMARY_HEADER]
    lines += [template.format(given[key]) for key, template in _SUMMARY_OPTIONAL if given[key] is not None]
    lines += ["", _SUMMARY_SECTION]
    epoch = len(hist["total"]["train"]) - 1
    for label, curves in [("TOTAL", hist["total"])] + [(cat, hist[cat]) for cat in categories]:
        at_epoch = {split: (curves[split][epoch] if (label == "TOTAL" or len(curves[split]) > 0) else float("nan"))
                    for split in ("train", "val")}
        lines.append(_SUMMARY_ROW.format(label=label, train=_six_decimals(at_epoch["train"]), val=_six_decimals(at_epoch["val"])))
    Path(path_txt).write_text("\n".join(lines) + "\n", encoding="utf-8")


def _plt():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    return plt


def plot_curve(xs, ys_dict, title, path):
    """Reference lines 86-99: every curve of ys_dict in one figure."""
    plt = _plt()
    plt.figure(figsize=(12, 8))
    for k in sorted(ys_dict.keys()):
        tr, va = ys_dict[k]
        plt.plot(xs, tr, label=f"{k}-Train")
        plt.plot(xs, va, "--", label=f"{k}-Val")
    plt.xlabel("Epoch"), plt.ylabel("KL Loss"), plt.title(title), plt.grid(True), plt.legend()
    plt.tight_layout(), plt.savefig(path), plt.close()


def plot_label_curve(xs, train_vals, val_vals, label_name, out_path):
    """Reference lines 101-112."""
    plot_pair(xs, train_vals, val_vals, ("Train", "Val"), f"{label_name} - KL Loss", out_path)


def plot_total_curve(xs, total_train, total_val, out_path):
    """Reference lines 114-125."""
    plot_pair(xs, total_train, total_val, ("Total-Train", "Total-Val"), "Overall KL Loss (Total)", out_path)


def plot_pair(xs, tr, va, names, title, out_path):
    plt = _plt()
    plt.figure(figsize=(10, 6))
    plt.plot(xs, tr, label=names[0])
    plt.plot(xs, va, "--", label=names[1])
    plt.xlabel("Epoch"), plt.ylabel("KL Loss"), plt.title(title), plt.grid(True), plt.legend()
    plt.tight_layout(), plt.savefig(out_path), plt.close()


def _criterion(out, batch):
    mu_pred, kappa_pred, w_pred = out
    return match_loss(mu_pred, kappa_pred, w_pred, batch[1], batch[1], batch[2])


_loss = (lambda model, batch: model(batch[0]), _criterion)    # (forward, criterion): the reference times them separately


def _dataset_loaders(rank, world):
    if not ROOT.exists():
        raise RuntimeError(f"ROOT not exists: {ROOT}")
    gt_txts = list(ROOT.rglob("*_multi_peak_vM_gt.txt"))
    if len(gt_txts) == 0:
        raise RuntimeError("No GT txts found under ROOT")
    categories = sorted(set(t.parent.name for t in gt_txts))
    label_map = {c: i for i, c in enumerate(categories)}
    samples = []
    for txt in gt_txts:
        cat = txt.parent.name
        ply_path = PLY_ROOT / cat / (txt.stem.replace("_multi_peak_vM_gt", "") + ".ply")
        if not ply_path.exists():
            raise FileNotFoundError(f"PLY not found: {ply_path}, for GT: {txt}")
        samples.append((str(ply_path), str(txt), cat))
    random.shuffle(samples)
    n_total = len(samples)
    n_tr, n_va = int(0.7 * n_total), int(0.15 * n_total)
    lo, hi = pdist.shard_bounds(n_tr, rank, world)
    parts = {"train": samples[:n_tr][lo:hi], "val": samples[n_tr:n_tr + n_va], "test": samples[n_tr + n_va:]}
    print(f"Samples: {n_total} | train:{n_tr} val:{n_va} test:{n_total - n_tr - n_va}")
    return categories, {k: DataLoader(PointCloudDatasetMvM(v, NUM_POINTS, max_K=4, label_map=label_map), BATCH,
                                      k == "train", num_workers=4, pin_memory=True) for k, v in parts.items()}


def _synthetic_loaders(n, rank):
    import synthetic
    out = {}
    g = torch.Generator().manual_seed(SEED + rank)
    for i, (name, frac) in enumerate((("train", 0.7), ("val", 0.15), ("test", 0.15))):
        m = max(BATCH, int(n * frac))
        xyz, _, _, fwd = synthetic.rotated_clouds(m, NUM_POINTS, seed=SEED + 1000 * i + rank)
        K = torch.tensor([1, 2, 4])[torch.randint(0, 3, (m,), generator=g)]
        out[name] = trainer.SyntheticLoader([xyz, synthetic.multi_peak_gt(fwd, K), K], BATCH, name == "train", device)
    return ["synthetic"], out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--sampler", default=os.environ.get("PNPP_SAMPLER", "randperm"), choices=["randperm", "device", "fps"])
    args = ap.parse_args(argv)
    rank, _, world = pdist.init_from_env()
    torch.manual_seed(SEED), np.random.seed(SEED), random.seed(SEED)
    sampling.reset(0)   # the device-side centre sampler restarts its stream too: a run is a function of SEED
    RES.mkdir(parents=True, exist_ok=True), FIGS.mkdir(parents=True, exist_ok=True)
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else device
    categories, loaders = _synthetic_loaders(args.synthetic, rank) if args.synthetic else _dataset_loaders(rank, world)
    model = PointNetPPMvM(sampler=args.sampler).to(dev)
    h, best_state, best_ep = trainer.fit(model, _loss, loaders, EPOCHS, LR, dev, clip_norm=1.0, label="multi-peak vM KL",
                                         label_index=3, n_labels=len(categories))
    hist = {"total": {"train": h["train"], "val": h["val"]}}
    for i, cat in enumerate(categories):
        hist[cat] = h["labels"][i]
    model.load_state_dict(best_state)
    test_kl = trainer.evaluate(model, _loss, loaders["test"], dev)
    if rank == 0:
        torch.save(best_state, RES / "mvM_best.pth")
        try:
            xs = list(range(1, EPOCHS + 1))
            for cat in categories:
                plot_label_curve(xs, hist[cat]["train"], hist[cat]["val"], cat, FIGS / f"loss_{_sanitize(cat)}.png")
            plot_total_curve(xs, hist["total"]["train"], hist["total"]["val"], FIGS / "loss_total.png")
            plot_curve(xs, {k: (v["train"], v["val"]) for k, v in hist.items()}, "Multi-Peak von Mises KL Loss", FIGS / "loss_overview.png")
        except Exception as e:  # plotting is optional (matplotlib may be absent)
            print(f"[plot skipped: {e}]")
        print(f"Test KL = {test_kl:.6f}  (steps: {h['steps']})")
        write_summary_txt(RES / "results.txt", categories, hist, test_kl=test_kl, best_val_epoch=best_ep)
    # beside the curves (the reference's `hist` layout): the trainer's own bookkeeping, for tools/script_throughput.py
    hist["_trainer"] = {"seconds": h["seconds"], "samples": h["samples"], "steps": h["steps"]}
    return hist, test_kl


if __name__ == "__main__":
    main()
