"""models/pointnet_pp_mvM.py -- drop-in for the reference's mixture-of-von-Mises model
(models/pointnet_pp_mvM.py:15-144): PointNet++ backbone, LayerNorm head, three output heads.

Parameter containers are created in the reference's order (sa1, sa2, sa3, fc1, ln1, fc2, ln2, head_pi, head_mu,
head_kappa), so state_dict keys and the seeded default initialisation are the reference's; the arithmetic runs on the
HIP kernels (pnpp_hip.ops)."""
import math

import torch
import torch.nn as nn

from pnpp_hip import ops, sampling
from .pointnet_pp_8dir import PointNetSetAbstraction, stacked_levels

_BACKBONE = ((128, 32, 0, (64, 64, 128), False), (32, 32, 128, (128, 128, 256), False), (None, None, 256, (256, 512, 1024), True))


def _as_points_last(xyz: torch.Tensor) -> torch.Tensor:
    """(B,N,3) or (B,3,N) -> (B,N,3).  Same acceptance rule and exception types as reference lines 15-27 (the last axis
    wins when both are 3)."""
    if xyz.dim() != 3:
        raise AssertionError(f"xyz should be 3D tensor, got {xyz.shape}")
    if xyz.shape[2] == 3:
        return xyz
    if xyz.shape[1] == 3:
        return xyz.transpose(1, 2)
    raise ValueError(f"xyz must be (B,N,3) or (B,3,N), got {xyz.shape}")


class PointNetPPMvM(nn.Module):
    """forward(xyz) -> mu (B,K) in [-pi,pi], kappa (B,K) >= 0, weight (B,K) summing to 1."""

    def __init__(self, max_K: int = 4, kappa_max: float = 80.0, p_drop: float = 0.4, temp: float = 0.7, sampler=None,
                 grouper=None):
        super().__init__()
        self.max_K, self.kappa_max, self.temp = max_K, float(kappa_max), float(temp)
        for i, (npoint, nsample, cin, mlp, whole) in enumerate(_BACKBONE, start=1):
            setattr(self, f"sa{i}", PointNetSetAbstraction(npoint, nsample, cin, list(mlp), group_all=whole,
                                                           **({} if whole else dict(sampler=sampler, grouper=grouper))))
        widths = (1024, 512, 256)
        for i in (1, 2):                                   # fc1, ln1, fc2, ln2 (reference lines 57-62)
            setattr(self, f"fc{i}", nn.Linear(widths[i - 1], widths[i]))
            setattr(self, f"ln{i}", nn.LayerNorm(widths[i]))
        self.drop = nn.Dropout(p_drop)
        for name, n_out in (("head_pi", max_K), ("head_mu", 2 * max_K), ("head_kappa", max_K)):
            setattr(self, name, nn.Linear(widths[-1], n_out))
        with torch.no_grad():                              # reference lines 69-73: uniform weights, undefined angles,
            for t in (self.head_pi.weight, self.head_pi.bias, self.head_mu.weight, self.head_mu.bias, self.head_kappa.bias):
                t.zero_()                                  # kappa = softplus(W x) at the start

    _presampled = None

    def use_presampled(self, ring) -> None:
        """ring: a pnpp_hip.sampling.CentreRing whose buffers hold the centres of the next forward pass (drawn one step ahead by
        the previous step's tail launch, `loss_backward(next_centres=ring.job())`); None: draw at the start of the step."""
        self._presampled = ring

    def _global_feat(self, pts: torch.Tensor, centres=None, drop_masks=(None, None)) -> torch.Tensor:
        B = pts.size(0)
        if centres is None and self._presampled is not None and self.training and self._presampled.B == B:
            centres = self._presampled.centres
        c1, c2 = centres if centres is not None else (None, None)
        if centres is None and self.sa1.sampler == "device" and self.sa2.sampler == "device":
            # neither draw depends on data: both levels' centres from one launch (same centres as the per-level draws)
            c1, c2 = sampling.device_random_centres_pair(B, pts.size(1), self.sa1.npoint, self.sa1.npoint, self.sa2.npoint, pts.device)
        _, _, l2_xyz, l2_pts = stacked_levels(self.sa1, self.sa2, pts, c1, c2)
        feat = self.sa3(l2_xyz, l2_pts)[1].flatten(1)
        for fc, ln, mask in ((self.fc1, self.ln1, drop_masks[0]), (self.fc2, self.ln2, drop_masks[1])):
            feat = ops.fc_block(feat, fc, ln, relu=True, dropout=self.drop, training=self.training, mask=mask)
        return feat

    def features(self, xyz: torch.Tensor, centres=None, drop_masks=(None, None)) -> torch.Tensor:
        """(B,N,3) | (B,3,N) -> (B,256): everything in front of the three output heads (reference lines 75-84)."""
        return self._global_feat(_as_points_last(xyz).contiguous(), centres, drop_masks)

    def forward(self, xyz: torch.Tensor, centres=None, drop_masks=(None, None)):
        feat = self.features(xyz, centres, drop_masks)
        raw = [ops.fc_block(feat, head, training=self.training) for head in (self.head_pi, self.head_mu, self.head_kappa)]
        # softmax(pi/temp); normalize(eps=1e-4) + degenerate fallback + atan2; softplus + 1e-6, clamp_max -- one kernel
        return ops.mvm_head(raw[0], raw[1], raw[2], self.temp, self.kappa_max)

    def loss_backward(self, xyz: torch.Tensor, vm_gt: torch.Tensor, K_gt: torch.Tensor, centres=None, drop_masks=(None, None),
                      next_centres=None, outputs=False):
        """forward(xyz) -> match_loss(mu, kappa, weight, vm_gt, K_gt).mean() -> loss.backward() of one training step
        (train_multi_peaks_vonMises_KL.py:224-234) with the three output heads, the head activations, the loss, its mean and their
        backward in ONE launch (ops.mvm_heads_match_loss_backward).  Returns the detached mean loss (and mu, kappa, weight with
        outputs=True); gradients are where loss.backward() would have put them."""
        feat = self.features(xyz, centres, drop_masks)
        return ops.mvm_heads_match_loss_backward(feat, self.head_pi, self.head_mu, self.head_kappa, vm_gt, K_gt, self.temp, self.kappa_max,
                                                 next_centres=next_centres, outputs=outputs)


@torch.no_grad()
def mvm_density_on_grid(mu, kappa, weight, num=360, device=None):
    """Mixture density on the first num-1 of `num` equispaced angles of [0, 2 pi], renormalised to sum 1 per sample
    (reference lines 130-144; an evaluation / plotting helper outside the training path, plain tensor ops).
    Returns (theta (num-1,), p (B,num-1))."""
    dev = device if device is not None else mu.device
    theta = torch.linspace(0.0, 2.0 * math.pi, steps=num, device=dev, dtype=mu.dtype)[:-1]
    log_vm = kappa.unsqueeze(-1) * torch.cos(theta - mu.unsqueeze(-1)) - torch.log(2.0 * math.pi * torch.i0(kappa)).unsqueeze(-1)
    p = torch.einsum("bk,bkt->bt", weight, log_vm.exp())
    return theta, p / (p.sum(dim=-1, keepdim=True) + 1e-8)
