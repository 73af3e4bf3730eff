"""models/pointnet_pp.py -- drop-in for the reference file of the same name (models/pointnet_pp.py:6-68):
PointNetSetAbstraction (the same block as in pointnet_pp_8dir) and PointNetPP, the backbone with a raw 3-vector head."""
import torch.nn as nn

from pnpp_hip import ops
from .pointnet_pp_8dir import BackboneBNHead, PointNetSetAbstraction  # noqa: F401  (the reference defines the block per file)


class PointNetPP(BackboneBNHead):
    """forward(x (B,N,3)) -> (B,3) raw direction vector (models/pointnet_pp.py:44-68)."""

    def __init__(self, sampler=None, grouper=None):
        super().__init__(sampler, grouper)
        self.fc3 = nn.Linear(256, 3)

    def forward(self, x, centres=None, drop_mask=None):
        return ops.fc_block(self.trunk(x, centres, drop_mask), self.fc3, training=self.training)
