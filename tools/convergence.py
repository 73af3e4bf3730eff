#!/usr/bin/env python3
"""Does the path LEARN?  Trains the drop-in single-peak von-Mises model and the 8-direction model with the drop-in trainer (hipGraph
step, device-side centre sampling, FlatAdam) on generated clouds and records the loss curves and, on held-out clouds, the angular
error of the predicted orientation.

The bench's synthetic clouds (a box rotated about +Y) are symmetric under a half turn, so their yaw is observable only
modulo 180 degrees and no single-peak predictor can do better than chance on the sign.  The clouds here are the same recipe with
the box tapered towards its front (a wedge: 0.4 of the width at the front, full width at the back), which makes the yaw observable.

The multi-peak model is not part of this: as the reference writes it (reproduced here), its mu head is zero-initialised, which
puts every component on the (c, s) = (1, 0) fallback whose gradient is zero (models/pointnet_pp_mvM.py:70-73,104-112), and
match_loss normalises by the weight of the first K components only (train_multi_peaks_vonMises_KL.py:77-79), so the optimiser
reaches KL = 0 within an epoch by moving the mixture weight to the unmatched components -- observed with the drop-in, and what
the reference's own code does on the same inputs.

    python tools/convergence.py [--clouds 8192] [--epochs 40] > profiles/<round>_convergence.json
"""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-pointcloud-orientation-estimation_amd"))

import torch  # noqa: E402


def wedge_clouds(B, N, seed):
    """-> xyz (B,N,3), mu (B,), forward axis (B,3): tapered box, random yaw about +Y, canonical forward axis (0,0,-1)."""
    g = torch.Generator().manual_seed(seed)
    p = (torch.rand(B, N, 3, generator=g) * 2 - 1) * torch.tensor([0.6, 0.3, 1.0])
    p[:, :, 0] *= 0.4 + 0.6 * (p[:, :, 2] + 1.0) / 2.0          # narrow at z = -1 (the front), full width at the back
    th = torch.rand(B, generator=g) * (2 * math.pi)
    c, s = torch.cos(th), torch.sin(th)
    R = torch.zeros(B, 3, 3)
    R[:, 0, 0], R[:, 0, 2], R[:, 1, 1], R[:, 2, 0], R[:, 2, 2] = c, s, 1.0, -s, c
    xyz = torch.einsum("bnj,bij->bni", p, R).contiguous()
    f = torch.einsum("bij,j->bi", R, torch.tensor([0.0, 0.0, -1.0]))
    return xyz.float(), torch.atan2(f[:, 0], -f[:, 2]).float(), f.float()


def ang_err_deg(a, b):
    d = (a - b + math.pi) % (2 * math.pi) - math.pi
    return d.abs() * 180.0 / math.pi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=8192)
    ap.add_argument("--epochs", type=int, default=40)
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16"], help="bf16: the opt-in bf16-operand MFMA mode")
    args = ap.parse_args()
    from models.pointnet_pp_8dir import DIRS_8, PointNetPP8Dir
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from pnpp_hip import ops, sampling, trainer
    import synthetic
    dev = torch.device("cuda", 0)
    ops.set_matmul_precision(args.precision)
    xyz_tr, mu_tr, f_tr = wedge_clouds(args.clouds, args.points, 1)
    xyz_va, mu_va, f_va = wedge_clouds(1024, args.points, 2)
    out = {"workload": f"tapered-box clouds, N={args.points}, batch {args.batch}, {args.clouds} training / 1024 held-out clouds, "
                       f"{args.epochs} epochs, Adam lr 1e-3, device-side centre sampling, hipGraph step, matmul precision {args.precision}",
           "runs": {}}

    # --- single-peak von Mises KL (train_single_peak_vonMises_KL.py's loss) ---
    torch.manual_seed(42)
    sampling.reset(0)
    model = PointNetPPVonMises(sampler="device").to(dev)
    kap = torch.full_like(mu_tr, 8.0)
    loaders = {"train": trainer.SyntheticLoader([xyz_tr, torch.stack([mu_tr, kap], 1)], args.batch, True, dev),
               "val": trainer.SyntheticLoader([xyz_va, torch.stack([mu_va, torch.full_like(mu_va, 8.0)], 1)], args.batch, False, dev)}
    loss = lambda m, b: ops.vm_head_kl_loss(m.features(b[0]), b[1][:, 0].contiguous(), b[1][:, 1].contiguous(), reduction="none")
    hist, best, best_ep = trainer.fit(model, loss, loaders, args.epochs, 1e-3, dev, label="von Mises KL", log=lambda *_: None)
    model.eval()

    def held_out():
        errs, kaps = [], []
        with torch.no_grad():
            for i in range(0, 1024, args.batch):
                mu, kappa = model(xyz_va[i:i + args.batch].to(dev))
                errs.append(ang_err_deg(mu.cpu(), mu_va[i:i + args.batch]))
                kaps.append(kappa.cpu())
        return torch.cat(errs), torch.cat(kaps)

    e, k = held_out()                 # the LAST epoch's model (its validation KL swings from epoch to epoch at this learning rate)
    model.load_state_dict(best)       # the checkpoint of the best validation epoch (trainer.fit keeps a copy)
    eb, kb = held_out()
    out["runs"]["single_peak_vonMises_KL"] = {
        "train_kl": [round(v, 4) for v in hist["train"]], "val_kl": [round(v, 4) for v in hist["val"]], "best_val_epoch": best_ep,
        "steps": hist["steps"], "train_seconds_per_epoch": round(sum(hist["seconds"]["train"]) / args.epochs, 3),
        "held_out_angular_error_deg": {"mean": round(float(e.mean()), 2), "median": round(float(e.median()), 2),
                                       "p90": round(float(e.kthvalue(int(0.9 * len(e))).values), 2)},
        "held_out_mean_kappa": round(float(k.mean()), 2),
        "best_checkpoint": {"held_out_angular_error_deg": {"mean": round(float(eb.mean()), 2), "median": round(float(eb.median()), 2),
                                                           "p90": round(float(eb.kthvalue(int(0.9 * len(eb))).values), 2)},
                            "held_out_mean_kappa": round(float(kb.mean()), 2)}}

    # --- 8-direction soft-label cross entropy (train_8dir_KL.py's loss) ---
    torch.manual_seed(42)
    sampling.reset(0)
    model8 = PointNetPP8Dir(sampler="device").to(dev)
    p_tr, p_va = synthetic.dir8_soft_labels(f_tr, DIRS_8), synthetic.dir8_soft_labels(f_va, DIRS_8)
    loaders = {"train": trainer.SyntheticLoader([xyz_tr, p_tr], args.batch, True, dev),
               "val": trainer.SyntheticLoader([xyz_va, p_va], args.batch, False, dev)}
    loss8 = lambda m, b: ops.soft_ce(m(b[0]), b[1])
    hist8, state8, best8 = trainer.fit(model8, loss8, loaders, args.epochs, 1e-3, dev, label="8-dir soft CE", log=lambda *_: None)
    model8.eval()

    def top1():
        hit = 0
        with torch.no_grad():
            for i in range(0, 1024, args.batch):
                logits = model8(xyz_va[i:i + args.batch].to(dev)).cpu()
                hit += int((logits.argmax(1) == p_va[i:i + args.batch].argmax(1)).sum())
        return hit

    hit = top1()
    model8.load_state_dict(state8)
    hit_best = top1()
    ent = float(-(p_va * torch.log(p_va.clamp_min(1e-12))).sum(1).mean())     # the soft labels' own entropy: the loss floor
    out["runs"]["8dir_soft_CE"] = {"train_ce": [round(v, 4) for v in hist8["train"]], "val_ce": [round(v, 4) for v in hist8["val"]],
                                   "best_val_epoch": best8, "label_entropy_floor": round(ent, 4),
                                   "held_out_top1_direction_accuracy": round(hit / 1024, 4),
                                   "best_checkpoint": {"held_out_top1_direction_accuracy": round(hit_best / 1024, 4)}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
