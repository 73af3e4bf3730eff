"""dataloader_8dir_sampled.py -- drop-in for the reference module of the same name (lines 18-57).

PointCloudDataset(samples, num_points, uniform_set, label_map=None):
    samples  [(ply_path, prob8_path, label_str), ...]
    item ->  (xyz (N,3), prob_gt (8,), label_idx int); classes in `uniform_set`, a missing or an unreadable
             `_8dir.txt` give the uniform distribution 0.125 (reference lines 46-55).
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from dataloader_common import read_ply, sample_pts  # noqa: F401


class PointCloudDataset(Dataset):
    def __init__(self, samples, num_points, uniform_set, label_map=None):
        self.samples = list(samples)
        self.num_points = num_points
        self.uniform_set = set(uniform_set)
        self.label2id = label_map or {}
        if not self.label2id:
            for _, _, lbl in self.samples:
                if lbl not in self.label2id:
                    self.label2id[lbl] = len(self.label2id)

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        ply_p, prob_p, lbl = self.samples[idx]
        xyz = torch.from_numpy(np.ascontiguousarray(sample_pts(read_ply(ply_p), self.num_points), dtype=np.float32))
        uniform = torch.full((8,), 0.125, dtype=torch.float32)
        if (lbl in self.uniform_set) or (not os.path.exists(prob_p)):
            prob = uniform
        else:
            try:
                arr = np.loadtxt(prob_p, dtype=np.float32).flatten()
                prob = torch.tensor(arr[:8], dtype=torch.float32)
            except Exception:
                prob = uniform
        return xyz, prob, self.label2id[lbl]
