#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""train_8dir_KL.py -- drop-in for the reference script of the same name: PointNetPP8Dir trained with the
soft-label cross entropy kl_loss_per_sample_from_logits (reference lines 60-68), here one fused HIP launch.
Outputs, as the reference writes them (lines 121-149): RES/8dir_KLdiv_0926.pth (best-validation weights),
FIGS/overall_loss.png, FIGS/<label>_loss.png, RES/summary.txt with one "<label>\t<test loss>" line per label and a final
"Overall\t<test loss>" line."""
import argparse
import os
import random

import numpy as np
import torch
from torch.utils.data import DataLoader

from dataloader_8dir_sampled import PointCloudDataset
from models.pointnet_pp_8dir import DIRS_8, PointNetPP8Dir
from pnpp_hip import sampling, dist as pdist, ops, trainer

ROOT = trainer.env_path("PNPP_ROOT", "/home/pablo/ForwardNet/data/2d_1to8_sampled")
RES = trainer.env_path("PNPP_RES", "/home/pablo/ForwardNet/results/8dir_KLdiv_0926")
FIGS = RES / "figs"
CKPT_NAME = "8dir_KLdiv_0926.pth"            # the reference's file name (line 122)
NUM_POINTS = int(os.environ.get("PNPP_NUM_POINTS", 10_000))
BATCH = int(os.environ.get("PNPP_BATCH", 16))
EPOCHS = int(os.environ.get("PNPP_EPOCHS", 200))
LR = float(os.environ.get("PNPP_LR", 1e-3))
SEED = int(os.environ.get("PNPP_SEED", 42))
UNIFORM_SET = set(filter(None, os.environ.get("PNPP_UNIFORM_SET", "bottle,plant,bowl").split(",")))
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def kl_loss_per_sample_from_logits(logits, p_target):
    """logits (B,8), p_target (B,8) soft labels -> (B,) cross entropy H(P,Q) = -sum P log softmax(logits)."""
    return ops.soft_ce(logits, p_target)


def plot_curve(xs, ys_dict, title, path):
    """Reference lines 30-37."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    path.parent.mkdir(parents=True, exist_ok=True)
    plt.figure()
    for k, (tr, va) in ys_dict.items():
        plt.plot(xs, tr, label=f"{k}-Tr")
        plt.plot(xs, va, ls="--", label=f"{k}-Val")
    plt.xlabel("Epoch"), plt.ylabel("KL (nats)"), plt.title(title)
    plt.grid(True), plt.legend(), plt.tight_layout(), plt.savefig(path), plt.close()


_loss = (lambda model, batch: model(batch[0]),                                   # forward
         lambda logits, batch: kl_loss_per_sample_from_logits(logits, batch[1]))  # criterion


def _dataset_loaders(rank, world):
    labels = sorted(d.name for d in ROOT.iterdir() if d.is_dir())
    label_map = {l: i for i, l in enumerate(labels)}
    samples = []
    for lbl in labels:
        for ply in (ROOT / lbl).glob("*.ply"):
            samples.append((ply, ply.with_name(ply.stem + "_8dir.txt"), lbl))
    random.shuffle(samples)
    n_total = len(samples)
    n_tr, n_va = int(0.7 * n_total), int(0.15 * n_total)
    lo, hi = pdist.shard_bounds(n_tr, rank, world)
    parts = {"train": samples[:n_tr][lo:hi], "val": samples[n_tr:n_tr + n_va], "test": samples[n_tr + n_va:]}
    print(f"Samples  train:{n_tr}  val:{n_va}  test:{n_total - n_tr - n_va}")
    return labels, {k: DataLoader(PointCloudDataset(v, NUM_POINTS, UNIFORM_SET, label_map), batch_size=BATCH, shuffle=k == "train",
                          num_workers=4, pin_memory=True) for k, v in parts.items()}


def _synthetic_loaders(n, rank):
    import synthetic
    out = {}
    for i, (name, frac) in enumerate((("train", 0.7), ("val", 0.15), ("test", 0.15))):
        m = max(BATCH, int(n * frac))
        xyz, _, _, fwd = synthetic.rotated_clouds(m, NUM_POINTS, seed=SEED + 1000 * i + rank)
        out[name] = trainer.SyntheticLoader([xyz, synthetic.dir8_soft_labels(fwd, DIRS_8)], BATCH, name == "train", device)
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
    labels, loaders = _synthetic_loaders(args.synthetic, rank) if args.synthetic else _dataset_loaders(rank, world)
    model = PointNetPP8Dir(sampler=args.sampler).to(dev)
    hist, best_state, best_ep = trainer.fit(model, _loss, loaders, EPOCHS, LR, dev, label="8-dir soft CE", label_index=2,
                                            n_labels=len(labels))
    model.load_state_dict(best_state)
    overall, per_label = trainer.evaluate(model, _loss, loaders["test"], dev, label_index=2, n_labels=len(labels))
    if rank == 0:
        torch.save(best_state, RES / CKPT_NAME)
        try:
            xs = range(1, EPOCHS + 1)
            plot_curve(xs, {"overall": (hist["train"], hist["val"])}, "Overall Loss (KL)", FIGS / "overall_loss.png")
            for i, l in enumerate(labels):
                plot_curve(xs, {l: (hist["labels"][i]["train"], hist["labels"][i]["val"])}, f"{l} Loss (KL)", FIGS / f"{l}_loss.png")
        except Exception as e:  # plotting is optional (matplotlib may be absent)
            print(f"[plot skipped: {e}]")
        with open(RES / "summary.txt", "w") as f:            # reference lines 147-149
            for l, v in zip(labels, per_label):
                f.write(f"{l}\t{v:.6f}\n")
            f.write(f"Overall\t{overall:.6f}\n")
        print(f"Test soft-CE = {overall:.6f}  (best val epoch {best_ep}; steps: {hist['steps']})")
    return hist, overall


if __name__ == "__main__":
    main()
