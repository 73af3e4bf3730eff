#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""simple_pointnet_train.py -- drop-in for the reference script of the same name (BASELINE configs[0]).

The reference trains its own `SimplePointNet` (simple_pointnet_train.py:86-113: shared per-point MLP 3-64-128-256 with
BatchNorm + ReLU, max over the points, Linear-BatchNorm-ReLU-Dropout(0.3), Linear to the 3 components of the forward
vector) with nn.MSELoss and Adam(lr 1e-3) (:242-244).  Here the same module tree (same parameter containers, so seeded
initialisation and state_dict are identical) runs on the HIP path: the per-point MLP and the max are one whole-cloud
set-abstraction call (`group_all` over raw coordinates: the conv -> BatchNorm -> ReLU chain on the MFMA GEMM kernels and
the split-K max pooling), the head is the fused Linear/BatchNorm/ReLU/Dropout launch, the loss the row-wise MSE kernel.

Same public names as the reference: read_ply, sample_points, PointCloudDataset, SimplePointNet, train_model,
test_model, main.  What the reference hard-codes is overridable here: PNPP_SIMPLE_ROOT (directory of *.ply + *.txt pairs),
PNPP_NUM_POINTS (10000), PNPP_BATCH (16), PNPP_EPOCHS (200), PNPP_LR, PNPP_SEED, PNPP_RES (where the curve is written);
`--synthetic M` trains on M generated rotated clouds whose target is their forward axis (no data set ships with the
reference).  `--points 256 --batch 4` is the configuration BASELINE.json quotes.
"""
import argparse
import os
import random
import time

import numpy as np
import torch
import torch.nn as nn

import dataloader_common as dc
from pnpp_hip import ops, optim, trainer

NUM_POINTS = int(os.environ.get("PNPP_NUM_POINTS", 10_000))      # reference :226
BATCH = int(os.environ.get("PNPP_BATCH", 16))                   # :231
EPOCHS = int(os.environ.get("PNPP_EPOCHS", 200))                # :246
LR = float(os.environ.get("PNPP_LR", 1e-3))                     # :244
SEED = int(os.environ.get("PNPP_SEED", 42))                     # :197
LABEL = os.environ.get("PNPP_SIMPLE_LABEL", "chair")            # :202
ROOT = trainer.env_path("PNPP_SIMPLE_ROOT", "/home/pablo/ForwardNet/data/modelnet40_normal_resampled_rotated_ply/" + LABEL)
RES = trainer.env_path("PNPP_RES", ".")


def read_ply(file_path):
    """ASCII PLY -> (n, 3+) array of the vertex rows (reference :18-31); parse failures raise RuntimeError."""
    try:
        return dc.read_ply(file_path)
    except (OSError, ValueError) as e:
        raise RuntimeError(f"cannot read point cloud {file_path}: {e}")


def sample_points(points, num_points=10000):
    """`num_points` rows, without replacement when there are enough, with replacement otherwise (reference :33-41)."""
    return dc.sample_pts(points, num_points)


class PointCloudDataset(torch.utils.data.Dataset):
    """(points (num_points,3) f32, forward vector (3,) f32) per `<name>.ply` + `<name>.txt` pair (reference :46-81)."""

    def __init__(self, rotated_dir, file_list, num_points=1024):
        self.rotated_dir, self.file_list, self.num_points = rotated_dir, list(file_list), num_points

    def __len__(self):
        return len(self.file_list)

    def __getitem__(self, idx):
        ply = os.path.join(self.rotated_dir, self.file_list[idx])
        cloud = np.ascontiguousarray(sample_points(read_ply(ply), self.num_points), dtype=np.float32)
        txt = ply.replace(".ply", ".txt")
        if not os.path.exists(txt):
            raise FileNotFoundError(f"forward-vector file missing: {txt}")
        with open(txt, "r") as f:
            fields = f.read().split()
        if len(fields) < 3:
            raise ValueError(f"forward-vector file has fewer than three numbers: {txt}")
        return torch.from_numpy(cloud), torch.tensor([float(v) for v in fields[:3]], dtype=torch.float32)


class SimplePointNet(nn.Module):
    """forward(x (B,N,3)) -> (B,3).  Parameter containers in the reference's construction order (:87-101)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv1d(3, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 256, 1)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(256)
        self.fc1 = nn.Linear(256, 128)
        self.bn4 = nn.BatchNorm1d(128)
        self.dropout = nn.Dropout(p=0.3)
        self.fc2 = nn.Linear(128, 3)

    def global_feature(self, x):
        """(B,N,3) -> (B,256): conv/bn/relu x 3 over every point, then the max over the cloud (:105-110)."""
        _, feat = ops.set_abstraction(x, None, None, None, True, self.training, (self.conv1, self.conv2, self.conv3),
                                      (self.bn1, self.bn2, self.bn3))
        return feat.view(x.size(0), -1)

    def forward(self, x, drop_mask=None):
        """drop_mask (B,128) of {0,1} replaces the dropout draw (parity runs)."""
        x = ops.fc_block(self.global_feature(x), self.fc1, self.bn4, relu=True, dropout=self.dropout, training=self.training,
                         mask=drop_mask)                                                   # :111-112
        return ops.fc_block(x, self.fc2, training=self.training)                           # :113


def criterion(outputs, target):
    """nn.MSELoss() of the reference (:243) as a per-sample vector; its mean is the reference's scalar."""
    return ops.mse_rows(outputs, target)


def train_model(model, criterion, optimizer, train_loader, val_loader, device, num_epochs=100):
    """The reference's loop (:118-163): returns (model with the best-validation weights, train losses, val losses).
    `criterion` may return the scalar batch loss or a per-sample vector; the running sums stay on the device and are read
    once per epoch.  `optimizer` is anything with zero_grad() / step() -- pnpp_hip.optim.FlatAdam in main()."""
    train_losses, val_losses = [], []
    best_val, best_state = float("inf"), None
    for epoch in range(num_epochs):
        sums = {}
        for phase, loader in (("train", train_loader), ("val", val_loader)):
            model.train() if phase == "train" else model.eval()
            total = torch.zeros((), device=device, dtype=torch.float64)
            count = 0
            for points, target in loader:
                points, target = points.to(device, non_blocking=True), target.to(device, non_blocking=True)
                if phase == "train":
                    optimizer.zero_grad()
                    loss = criterion(model(points), target).mean()
                    loss.backward()
                    optimizer.step()
                else:
                    with torch.no_grad():
                        loss = criterion(model(points), target).mean()
                total += loss.detach().double() * points.size(0)
                count += points.size(0)
            sums[phase] = float(total) / max(count, 1)
        train_losses.append(sums["train"])
        val_losses.append(sums["val"])
        print(f"Epoch [{epoch + 1}/{num_epochs}] Train Loss: {sums['train']:.6f}, Val Loss: {sums['val']:.6f}")
        if sums["val"] < best_val:
            best_val = sums["val"]
            best_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    if best_state is not None:
        model.load_state_dict(best_state)
    return model, train_losses, val_losses


def test_model(model, test_loader, criterion, device, verbose=True):
    """Mean test loss; prints the per-sample and per-batch lines the reference prints (:165-191)."""
    model.eval()
    total, count = 0.0, 0
    with torch.no_grad():
        for points, target in test_loader:
            points, target = points.to(device), target.to(device)
            outputs = model(points)
            rows = ops.mse_rows(outputs, target).cpu()
            batch_loss = float(criterion(outputs, target).mean()) if criterion is not None else float(rows.mean())
            if verbose:
                for v in rows.tolist():
                    print(f"Manual MSE loss for this sample: {v}")
                print(f"Manual computed batch MSE loss: {float(rows.mean())}")
                print(f"Criterion computed batch MSE loss: {batch_loss}")
            total += batch_loss * points.size(0)
            count += points.size(0)
    test_loss = total / max(count, 1)
    print(f"Test Loss: {test_loss:.6f}")
    return test_loss


def _plot(train_losses, val_losses, label, path):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    xs = range(1, len(train_losses) + 1)
    plt.figure()
    plt.plot(xs, train_losses, label="Train Loss")
    plt.plot(xs, val_losses, label="Validation Loss")
    plt.xlabel("Epoch"), plt.ylabel("MSE Loss"), plt.title(f"{label} Training and Validation Loss")
    plt.legend(), plt.grid(True), plt.savefig(path), plt.close()


def _file_loaders(num_points, batch):
    files = sorted(f for f in os.listdir(ROOT) if f.endswith(".ply"))
    n = len(files)
    n_tr, n_va = int(0.7 * n), int(0.15 * n)
    random.shuffle(files)
    parts = {"train": files[:n_tr], "val": files[n_tr:n_tr + n_va], "test": files[n_tr + n_va:]}
    print(f"samples: {n}, train: {len(parts['train'])}, val: {len(parts['val'])}, test: {len(parts['test'])}")
    return {k: torch.utils.data.DataLoader(PointCloudDataset(str(ROOT), v, num_points), batch_size=batch, shuffle=k == "train",
                                           num_workers=4) for k, v in parts.items()}


class _Pairs:
    """(points, target) view of a SyntheticLoader, whose batches carry a label column last."""

    def __init__(self, loader):
        self.loader = loader

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        return (b[:2] for b in self.loader)


def _synthetic_loaders(m, num_points, batch, device):
    import synthetic
    out = {}
    for i, (name, frac) in enumerate((("train", 0.7), ("val", 0.15), ("test", 0.15))):
        xyz, _, _, fwd = synthetic.rotated_clouds(max(batch, int(m * frac)), num_points, seed=SEED + 1000 * i)
        out[name] = _Pairs(trainer.SyntheticLoader([xyz, fwd.float()], batch, name == "train", device))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic", type=int, default=0, help="train on this many generated clouds instead of PNPP_SIMPLE_ROOT")
    ap.add_argument("--points", type=int, default=NUM_POINTS)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--epochs", type=int, default=EPOCHS)
    ap.add_argument("--quiet", action="store_true", help="no per-sample lines in the test phase")
    args = ap.parse_args(argv)
    random.seed(SEED), np.random.seed(SEED), torch.manual_seed(SEED)
    if not torch.cuda.is_available():
        raise RuntimeError("simple_pointnet_train: no GPU -- this drop-in runs on the HIP kernels only")
    device = torch.device("cuda", torch.cuda.current_device())
    print(f"device: {device}")
    loaders = (_synthetic_loaders(args.synthetic, args.points, args.batch, device) if args.synthetic
               else _file_loaders(args.points, args.batch))
    model = SimplePointNet().to(device)
    optimizer = optim.FlatAdam(model.parameters(), lr=LR)
    t0 = time.time()
    model, tr, va = train_model(model, criterion, optimizer, loaders["train"], loaders["val"], device, args.epochs)
    print(f"training took {(time.time() - t0) / 60:.2f} min")
    test_loss = test_model(model, loaders["test"], criterion, device, verbose=not args.quiet)
    try:
        RES.mkdir(parents=True, exist_ok=True)
        _plot(tr, va, LABEL, RES / f"{LABEL}_simplepointnet_training_validation_loss.png")      # reference :268
    except Exception as e:
        print(f"[plot skipped: {e}]")
    return tr, va, test_loss


if __name__ == "__main__":
    main()
