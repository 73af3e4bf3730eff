"""dataloader_single_peak_vonMises.py -- drop-in for the reference module of the same name (lines 5-52).

PointCloudDatasetVonMises(samples, num_points, label_map=None):
    samples  [(ply_path, label_str), ...]
    item ->  (xyz FloatTensor (N,3), vm_params FloatTensor (2,) = [mu, kappa], label_idx int)
The ground truth is read from `<stem>_single_peak_vM_gt.txt` next to the PLY: first non-comment line, first two
numbers; any failure silently yields (0, 0) and kappa is clamped at 0 -- the reference's behaviour (lines 36-45).
"""
from pathlib import Path

import torch
from torch.utils.data import Dataset

from dataloader_common import LabelledPlyDataset, read_ply, sample_pts  # noqa: F401  (read_ply / sample_pts: names of the reference module)


def parse_single_peak_gt(path):
    """-> (mu, kappa >= 0); (0, 0) for a missing, empty or malformed file."""
    try:
        rows = [ln.split() for ln in Path(path).read_text(encoding="utf-8").splitlines()
                if ln.strip() and not ln.startswith("#")]
        mu, kappa = float(rows[0][0]), float(rows[0][1])
    except Exception:
        return 0.0, 0.0
    return mu, max(kappa, 0.0)


class PointCloudDatasetVonMises(LabelledPlyDataset, Dataset):
    def ground_truth(self, sample):
        ply = Path(sample[0])
        return (torch.tensor(parse_single_peak_gt(ply.with_name(ply.stem + "_single_peak_vM_gt.txt")), dtype=torch.float32),)
