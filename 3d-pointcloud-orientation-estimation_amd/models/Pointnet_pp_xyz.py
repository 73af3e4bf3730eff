"""models/Pointnet_pp_xyz.py -- drop-in for the reference file of the same name (models/Pointnet_pp_xyz.py:6-90):
PointNetPPXYZ, the backbone with two unit-vector heads (rotated X and Y axes)."""
import torch.nn as nn

from pnpp_hip import ops
from .pointnet_pp_8dir import BackboneBNHead, PointNetSetAbstraction  # noqa: F401


class PointNetPPXYZ(BackboneBNHead):
    """forward(x (B,N,3)) -> (v1, v2), each (B,3), L2-normalised head_x / head_y outputs (lines 47-90)."""

    def __init__(self, sampler=None, grouper=None):
        super().__init__(sampler, grouper)
        self.head_x = nn.Linear(256, 3)
        self.head_y = nn.Linear(256, 3)

    def forward(self, x, centres=None, drop_mask=None):
        feat = self.trunk(x, centres, drop_mask)
        v1 = ops.l2_normalize(ops.fc_block(feat, self.head_x, training=self.training))
        v2 = ops.l2_normalize(ops.fc_block(feat, self.head_y, training=self.training))
        return v1, v2
