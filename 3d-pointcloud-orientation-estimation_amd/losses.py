"""Loss expressions of the reference's direction-vector training scripts on the HIP kernels.

    axis_pair_loss   train.py:183-187        (MSE(vy,gy) + MSE(vz,gz)) / 2 + lam * mean((vy . vz)^2)
    proj_probs       train_multi_8dir.py:41-44, train_8dir_MSE.py (same helper)
    mse_loss         nn.MSELoss()            train.py:168, train_8dir.py:53, train_multi_8dir.py:80
"""
import torch

from models.pointnet_pp_8dir import DIRS_8
from pnpp_hip import ops

mse_loss = ops.mse_loss


def axis_pair_loss(vy, vz, gy, gz, lam: float = 0.1):
    """train.py:183-187 (lam is the reference's hard-coded 0.1)."""
    pred_loss = (ops.mse_loss(vy, gy) + ops.mse_loss(vz, gz)) / 2.0
    return pred_loss + lam * ops.orth_loss(vy, vz)


_dirs_cache = {}


def proj_probs(vec: torch.Tensor) -> torch.Tensor:
    """vec (B,3) -> (B,8) probabilities over DIRS_8 (train_multi_8dir.py:41-44)."""
    d = _dirs_cache.get(vec.device)
    if d is None:
        d = DIRS_8.to(vec.device)
        _dirs_cache[vec.device] = d
    return ops.proj_probs(vec, d)
