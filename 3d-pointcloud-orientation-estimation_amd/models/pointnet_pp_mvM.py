"""models/pointnet_pp_mvM.py -- drop-in for the reference's mixture-of-von-Mises model
(models/pointnet_pp_mvM.py:15-144): PointNet++ backbone, LayerNorm head, three output heads."""
import math

import torch
import torch.nn as nn

from pnpp_hip import ops
from .pointnet_pp_8dir import PointNetSetAbstraction


def _maybe_transpose_xyz(xyz: torch.Tensor) -> torch.Tensor:
    """Returns xyz as (B,3,N); accepts (B,N,3) or (B,3,N) (reference lines 15-27, same checks)."""
    assert xyz.dim() == 3, f"xyz should be 3D tensor, got {xyz.shape}"
    B, A, C = xyz.shape
    if C == 3:
        xyz = xyz.transpose(1, 2).contiguous()
    elif A == 3:
        pass
    else:
        raise ValueError(f"xyz must be (B,N,3) or (B,3,N), got {xyz.shape}")
    return xyz


class PointNetPPMvM(nn.Module):
    """forward(xyz) -> mu (B,K) in [-pi,pi], kappa (B,K) >= 0, weight (B,K) summing to 1."""

    def __init__(self, max_K: int = 4, kappa_max: float = 80.0, p_drop: float = 0.4, temp: float = 0.7):
        super().__init__()
        self.max_K = max_K
        self.kappa_max = float(kappa_max)
        self.temp = float(temp)

        self.sa1 = PointNetSetAbstraction(128, 32, 0, [64, 64, 128])
        self.sa2 = PointNetSetAbstraction(32, 32, 128, [128, 128, 256])
        self.sa3 = PointNetSetAbstraction(None, None, 256, [256, 512, 1024], group_all=True)

        self.fc1 = nn.Linear(1024, 512)
        self.ln1 = nn.LayerNorm(512)
        self.fc2 = nn.Linear(512, 256)
        self.ln2 = nn.LayerNorm(256)
        self.drop = nn.Dropout(p_drop)

        hidden = 256
        self.head_pi = nn.Linear(hidden, max_K)
        self.head_mu = nn.Linear(hidden, max_K * 2)
        self.head_kappa = nn.Linear(hidden, max_K)

        nn.init.zeros_(self.head_pi.weight)
        nn.init.zeros_(self.head_pi.bias)
        nn.init.zeros_(self.head_mu.weight)
        nn.init.zeros_(self.head_mu.bias)
        nn.init.constant_(self.head_kappa.bias, 0.0)

    def _global_feat(self, xyz: torch.Tensor, centres=None, drop_masks=(None, None)) -> torch.Tensor:
        B = xyz.size(0)
        xyz_bn3 = xyz.transpose(1, 2).contiguous()
        c1, c2 = centres if centres is not None else (None, None)
        l1_xyz, l1_pts = self.sa1(xyz_bn3, None, c1)
        l2_xyz, l2_pts = self.sa2(l1_xyz, l1_pts, c2)
        _, l3_pts = self.sa3(l2_xyz, l2_pts)
        x = l3_pts.view(B, -1)
        x = ops.fc_block(x, self.fc1, self.ln1, relu=True, dropout=self.drop, training=self.training, mask=drop_masks[0])
        x = ops.fc_block(x, self.fc2, self.ln2, relu=True, dropout=self.drop, training=self.training, mask=drop_masks[1])
        return x

    def forward(self, xyz: torch.Tensor, centres=None, drop_masks=(None, None)):
        xyz = _maybe_transpose_xyz(xyz)
        feat = self._global_feat(xyz, centres, drop_masks)
        pi_raw = ops.fc_block(feat, self.head_pi, training=self.training)
        mu_raw = ops.fc_block(feat, self.head_mu, training=self.training)
        kappa_raw = ops.fc_block(feat, self.head_kappa, training=self.training)
        # softmax(pi/temp); normalize(eps=1e-4) + degenerate fallback + atan2; softplus + 1e-6, clamp_max
        return ops.mvm_head(pi_raw, mu_raw, kappa_raw, self.temp, self.kappa_max)


@torch.no_grad()
def mvm_density_on_grid(mu, kappa, weight, num=360, device=None):
    """Mixture density sampled on num-1 angles of [0, 2pi) (reference lines 130-144; evaluation /
    plotting helper outside the training path, plain tensor ops)."""
    B, K = mu.shape
    device = device or mu.device
    theta = torch.linspace(0.0, 2 * math.pi, steps=num, device=device, dtype=mu.dtype)[:-1][None, None, :]
    mu, kappa, w = mu[..., None], kappa[..., None], weight[..., None]
    vm = torch.exp(kappa * torch.cos(theta - mu)) / (2 * math.pi * torch.i0(kappa))
    p = (w * vm).sum(dim=1)
    p = p / (p.sum(dim=-1, keepdim=True) + 1e-8)
    return theta.squeeze(), p
