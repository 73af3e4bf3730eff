"""Synthetic orientation workload: random-rotated N-point clouds with their ground truth.

Recipe, from the reference's offline data scripts (no dataset ships with the reference):
  * base cloud: uniform in an anisotropic box (1.0, 0.6, 0.3) so that yaw is observable; continuous
    coordinates, hence tie-free neighbour distances;
  * random yaw about +Y: R = [[c,0,s],[0,1,0],[-s,0,c]], xyz = p R^T   (data_process/rotate_without_normals.py:5-15,112);
  * forward axis f = R (0,0,-1); mu = atan2(f_x, -f_z)                  (data_process/2d_multi_peak_MvM_gt_1.py:50-59);
  * kappa = 8                                                           (data_process/2d_single_peak_vM_gt.py:8);
  * multi-peak GT: first K of [front, -front, side, -side], kappa 8, weight 1/K, zero padded to max_K
    (2d_multi_peak_MvM_gt_1.py:66-72, dataloader_multi_peak_vonMises.py:59-63);
  * 8-direction soft labels: relu(DIRS_8 . f) normalised               (data_process/2d_8dir_sample.py:32-39).
Generated on the CPU generator (deterministic for a seed), returned as CPU tensors.
"""
import math

import torch


def rotated_clouds(B: int, N: int, seed: int = 1234):
    """-> xyz (B,N,3) f32, mu (B,) f32, kappa (B,) f32, forward axis (B,3) f32."""
    g = torch.Generator().manual_seed(seed)
    p = (torch.rand(B, N, 3, generator=g) * 2 - 1) * torch.tensor([1.0, 0.6, 0.3])
    th = torch.rand(B, generator=g) * (2 * math.pi)
    c, s = torch.cos(th), torch.sin(th)
    R = torch.zeros(B, 3, 3)
    R[:, 0, 0], R[:, 0, 2], R[:, 1, 1], R[:, 2, 0], R[:, 2, 2] = c, s, 1.0, -s, c
    xyz = torch.einsum("bnj,bij->bni", p, R).contiguous()
    f = torch.einsum("bij,j->bi", R, torch.tensor([0.0, 0.0, -1.0]))
    mu = torch.atan2(f[:, 0], -f[:, 2])
    return xyz.float(), mu.float(), torch.full((B,), 8.0), f


def multi_peak_gt(fwd: torch.Tensor, K: torch.Tensor, max_K: int = 4, kappa: float = 8.0):
    """-> vm_gt (B,max_K,3) rows [mu, kappa, weight] for the first K[b] peaks, zeros beyond."""
    B = fwd.shape[0]
    side = torch.stack([-fwd[:, 2], torch.zeros(B), fwd[:, 0]], 1)
    peaks = torch.stack([fwd, -fwd, side, -side], 1)
    vm = torch.zeros(B, max_K, 3)
    for b in range(B):
        k = int(K[b])
        for j in range(min(k, max_K)):
            vm[b, j] = torch.tensor([math.atan2(float(peaks[b, j, 0]), -float(peaks[b, j, 2])), kappa, 1.0 / k])
    return vm


def dir8_soft_labels(fwd: torch.Tensor, dirs8: torch.Tensor):
    p = torch.relu(fwd @ dirs8.t())
    return p / p.sum(1, keepdim=True)
