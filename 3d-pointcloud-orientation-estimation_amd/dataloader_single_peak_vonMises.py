"""dataloader_single_peak_vonMises.py -- drop-in for the reference module of the same name (lines 5-52).

PointCloudDatasetVonMises(samples, num_points, label_map=None):
    samples  [(ply_path, label_str), ...]
    item ->  (xyz FloatTensor (N,3), vm_params FloatTensor (2,) = [mu, kappa], label_idx int)
The ground truth is read from `<stem>_single_peak_vM_gt.txt` next to the PLY: first non-comment line, first two
numbers; any failure silently yields (0, 0) and kappa is clamped at 0 -- the reference's behaviour (lines 36-45).
"""
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset

from dataloader_common import read_ply, sample_pts  # noqa: F401  (re-exported names of the reference module)


class PointCloudDatasetVonMises(Dataset):
    def __init__(self, samples, num_points, label_map=None):
        self.samples = list(samples)
        self.num_points = num_points
        self.label2id = label_map or {}
        if not self.label2id:
            for _, lbl in self.samples:
                if lbl not in self.label2id:
                    self.label2id[lbl] = len(self.label2id)

    def __len__(self):
        return len(self.samples)

    @staticmethod
    def _read_vm(path):
        try:
            with open(path, "r", encoding="utf-8") as f:
                lines = [ln.strip() for ln in f if ln.strip() and not ln.startswith("#")]
            mu, kappa = map(float, lines[0].split()[:2])
        except Exception:
            mu, kappa = 0.0, 0.0
        return mu, max(kappa, 0.0)

    def __getitem__(self, idx):
        ply_p, lbl = self.samples[idx]
        ply_p = Path(ply_p)
        xyz = torch.from_numpy(np.ascontiguousarray(sample_pts(read_ply(ply_p), self.num_points), dtype=np.float32))
        mu, kappa = self._read_vm(ply_p.with_name(ply_p.stem + "_single_peak_vM_gt.txt"))
        return xyz, torch.tensor([mu, kappa], dtype=torch.float32), self.label2id[lbl]
