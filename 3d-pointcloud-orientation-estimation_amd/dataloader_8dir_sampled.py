"""dataloader_8dir_sampled.py -- drop-in for the reference module of the same name (lines 18-57).

PointCloudDataset(samples, num_points, uniform_set, label_map=None):
    samples  [(ply_path, prob8_path, label_str), ...]
    item ->  (xyz (N,3), prob_gt (8,), label_idx int); classes in `uniform_set`, a missing or an unreadable
             `_8dir.txt` give the uniform distribution 0.125 (reference lines 46-55).
"""
import numpy as np
import torch
from torch.utils.data import Dataset

from dataloader_common import LabelledPlyDataset, read_ply, sample_pts  # noqa: F401


def parse_8dir_probs(path):
    """First eight numbers of the soft-label file as float32, or None when it cannot be read."""
    try:
        return torch.from_numpy(np.loadtxt(path, dtype=np.float32).reshape(-1)[:8].copy())
    except Exception:
        return None


class PointCloudDataset(LabelledPlyDataset, Dataset):
    def __init__(self, samples, num_points, uniform_set, label_map=None):
        super().__init__(samples, num_points, label_map)
        self.uniform_set = set(uniform_set)

    def ground_truth(self, sample):
        prob = None if sample[-1] in self.uniform_set else parse_8dir_probs(sample[1])
        return (prob if prob is not None else torch.full((8,), 0.125, dtype=torch.float32),)
