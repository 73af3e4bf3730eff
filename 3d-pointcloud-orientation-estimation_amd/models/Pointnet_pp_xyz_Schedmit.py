"""models/Pointnet_pp_xyz_Schedmit.py -- drop-in for the reference file of the same name (lines 6-111):
PointNetPPXYZ_Schedmit, the backbone with two unit-vector heads (rotated Y = upright and Z = forward axes)."""
import torch.nn as nn

from pnpp_hip import ops
from .pointnet_pp_8dir import BackboneBNHead, PointNetSetAbstraction  # noqa: F401


class PointNetPPXYZ_Schedmit(BackboneBNHead):
    """forward(x (B,N,3)) -> (v2, v3), each (B,3), L2-normalised head_y / head_z outputs (lines 47-90)."""

    def __init__(self, sampler=None, grouper=None):
        super().__init__(sampler, grouper)
        self.head_y = nn.Linear(256, 3)
        self.head_z = nn.Linear(256, 3)

    def forward(self, x, centres=None, drop_mask=None):
        feat = self.trunk(x, centres, drop_mask)
        v2 = ops.l2_normalize(ops.fc_block(feat, self.head_y, training=self.training))
        v3 = ops.l2_normalize(ops.fc_block(feat, self.head_z, training=self.training))
        return v2, v3
