"""dataloader_multi_peak_vonMises.py -- drop-in for the reference module of the same name (lines 28-86).

PointCloudDatasetMvM(samples, num_points, max_K=4, label_map=None):
    samples  [(ply_path, gt_txt_path, category), ...]
    item ->  (xyz (N,3), vm_params (max_K,3) rows [mu, kappa, weight] zero padded, K int, label LongTensor scalar)
GT text: first non-comment line "K <int>", second line a header, then one "mu kappa weight" row per peak
(reference lines 36-64).  Missing files raise FileNotFoundError, malformed ones RuntimeError, as in the reference.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from dataloader_common import LabelledPlyDataset, read_ply, sample_pts  # noqa: F401


class PointCloudDatasetMvM(LabelledPlyDataset, Dataset):
    sorted_labels = True           # reference line 33: categories numbered in sorted order when no map is given

    def __init__(self, samples, num_points, max_K=4, label_map=None):
        super().__init__(samples, num_points, label_map)
        self.max_K = max_K
        self.label_map = self.label2id      # the reference's attribute name in this class

    @staticmethod
    def _read_mvM(gt_path, max_K=4):
        """-> (table (max_K,3) float32 [mu, kappa, weight] zero padded / truncated, K as written in the file)."""
        with open(gt_path, "r", encoding="utf-8") as f:
            lines = [ln.strip() for ln in f if ln.strip() and not ln.startswith("#")]
        if len(lines) < 2:
            raise RuntimeError(f"GT file too short or malformed: {gt_path}")
        head = lines[0].split()
        if len(head) < 2:
            raise RuntimeError(f"GT file K line malformed: {gt_path}")
        peaks = [tok[:3] for tok in (ln.split() for ln in lines[2:]) if len(tok) >= 3][:max_K]   # lines[1] is the column header
        table = np.zeros((max_K, 3), dtype=np.float32)
        if peaks:
            table[:len(peaks)] = np.asarray(peaks, dtype=np.float64)
        return torch.from_numpy(table), int(head[1])

    def cloud(self, ply_path):
        if not os.path.exists(ply_path):
            raise FileNotFoundError(f"PLY not found: {ply_path}")
        return super().cloud(ply_path)

    def ground_truth(self, sample):
        ply_p, gt_txt = sample[0], sample[1]
        if not os.path.exists(gt_txt):
            raise FileNotFoundError(f"GT txt not found: {gt_txt}, for ply: {ply_p}")
        return self._read_mvM(gt_txt, self.max_K)

    def label(self, sample):
        return torch.tensor(self.label2id[sample[-1]], dtype=torch.long)
