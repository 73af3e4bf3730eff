#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""train_single_peak_vonMises_KL.py -- drop-in for the reference script of the same name.

Same module-level names (ROOT, RES, FIGS, NUM_POINTS, BATCH, EPOCHS, LR, SEED, device, kl_von_mises, plot_curve),
same dataset layout, same outputs (RES/vonMises_best.pth, FIGS/loss.png, "Test KL = ..." line); the model, the
loss and the optimiser step run on the MI355X HIP kernels.  Differences from the reference, all deliberate:
  * paths and hyper-parameters are overridable (environment or command line) instead of hard-coded;
  * importing the module does not start training (the reference trains at import time, line 39 onwards);
  * `--synthetic N` trains on N generated clouds (synthetic.py) because no dataset ships with the reference;
  * under torchrun every rank trains on its own shard and gradients are all-reduced over RCCL.
"""
import argparse
import os
import random
import sys
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader

from dataloader_single_peak_vonMises import PointCloudDatasetVonMises
from models.pointnet_pp_vonMises import PointNetPPVonMises
from pnpp_hip import sampling, dist as pdist, ops, trainer

ROOT = trainer.env_path("PNPP_ROOT", "/home/pablo/ForwardNet/data/chair_toilet_sofa_plant_bowl_bottle")
RES = trainer.env_path("PNPP_RES", "/home/pablo/ForwardNet/results/single_peak_vonMises_KL_1006_2")
FIGS = RES / "figs"

NUM_POINTS = int(os.environ.get("PNPP_NUM_POINTS", 10_000))
BATCH = int(os.environ.get("PNPP_BATCH", 16))
EPOCHS = int(os.environ.get("PNPP_EPOCHS", 200))
LR = float(os.environ.get("PNPP_LR", 1e-3))
SEED = int(os.environ.get("PNPP_SEED", 42))
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def kl_von_mises(mu_p, kappa_p, mu_q, kappa_q):
    """KL( vM(mu_p, kappa_p) || vM(mu_q, kappa_q) ) per sample, p = prediction, q = ground truth
    (reference lines 23-28): value and analytic gradient from one fused HIP launch."""
    return ops.kl_von_mises_single(mu_p, kappa_p, mu_q, kappa_q)


def plot_curve(xs, ys_dict, title, path):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    plt.figure()
    for k, (tr, va) in ys_dict.items():
        plt.plot(xs, tr, label=f"{k}-Tr")
        plt.plot(xs, va, "--", label=f"{k}-Val")
    plt.xlabel("Epoch"), plt.ylabel("KL"), plt.title(title), plt.grid(True), plt.legend()
    plt.tight_layout()
    plt.savefig(path)
    plt.close()


def _loss(model, batch):
    """model.features = the raw fc3 output; head (tanh * pi, softplus), KL and their gradient are one launch
    (models/pointnet_pp_vonMises.py:36-37 + kl_von_mises): what `kl_von_mises(*model(xyz), mu_gt, kappa_gt)` computes."""
    xyz, vm_gt = batch[0], batch[1]
    return ops.vm_head_kl_loss(model.features(xyz), vm_gt[:, 0].contiguous(), vm_gt[:, 1].contiguous(), reduction="none")


def _dataset_loaders(rank, world):
    labels = sorted(d.name for d in ROOT.iterdir() if d.is_dir())
    label_map = {l: i for i, l in enumerate(labels)}
    samples = []
    for lbl in labels:
        for vm_file in (ROOT / lbl).glob("*_single_peak_vM_gt.txt"):
            ply_path = vm_file.with_name(vm_file.name.replace("_single_peak_vM_gt.txt", ".ply"))
            if ply_path.exists():
                samples.append((ply_path, lbl))
    random.shuffle(samples)
    n_total = len(samples)
    n_tr, n_va = int(0.7 * n_total), int(0.15 * n_total)
    lo, hi = pdist.shard_bounds(n_tr, rank, world)
    parts = {"train": samples[:n_tr][lo:hi], "val": samples[n_tr:n_tr + n_va], "test": samples[n_tr + n_va:]}
    print(f"Samples found: {n_total} | train:{n_tr} val:{n_va} test:{n_total - n_tr - n_va}")
    return {k: DataLoader(PointCloudDatasetVonMises(v, NUM_POINTS, label_map), BATCH, k == "train", num_workers=4,
                          pin_memory=True) for k, v in parts.items()}


def _synthetic_loaders(n, rank):
    import synthetic
    out = {}
    for i, (name, frac) in enumerate((("train", 0.7), ("val", 0.15), ("test", 0.15))):
        m = max(BATCH, int(n * frac))
        xyz, mu, kappa, _ = synthetic.rotated_clouds(m, NUM_POINTS, seed=SEED + 1000 * i + rank)
        out[name] = trainer.SyntheticLoader([xyz, torch.stack([mu, kappa], 1)], BATCH, name == "train", device)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic", type=int, default=0, help="train on this many generated clouds instead of ROOT")
    ap.add_argument("--sampler", default=os.environ.get("PNPP_SAMPLER", "randperm"), choices=["randperm", "device", "fps"])
    args = ap.parse_args(argv)
    rank, _, world = pdist.init_from_env()
    torch.manual_seed(SEED), np.random.seed(SEED), random.seed(SEED)
    sampling.reset(0)   # the device-side centre sampler restarts its stream too: a run is a function of SEED
    RES.mkdir(parents=True, exist_ok=True), FIGS.mkdir(parents=True, exist_ok=True)
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else device
    loaders = _synthetic_loaders(args.synthetic, rank) if args.synthetic else _dataset_loaders(rank, world)
    model = PointNetPPVonMises(sampler=args.sampler).to(dev)
    hist, best_state, _ = trainer.fit(model, _loss, loaders, EPOCHS, LR, dev, label="von Mises KL")
    if rank == 0:
        torch.save(best_state, RES / "vonMises_best.pth")
        try:
            plot_curve(range(1, EPOCHS + 1), {"KL": (hist["train"], hist["val"])}, "von Mises KL", FIGS / "loss.png")
        except Exception as e:  # plotting is optional
            print(f"[plot skipped: {e}]")
    model.load_state_dict(best_state)
    test_kl = trainer.evaluate(model, _loss, loaders["test"], dev)
    if rank == 0:
        print(f"Test KL = {test_kl:.6f}")
        print(f"[steps: {hist['steps']}]")
    return hist, test_kl


if __name__ == "__main__":
    main()
