"""models/pointnet_pp_Fwd.py -- drop-in for the reference file of the same name (models/pointnet_pp_Fwd.py:5-98):
PointNetSetAbstraction, DIRS_8 and PointNetPPFwd, the backbone with a unit 3-vector (forward axis) head."""
import torch.nn as nn

from pnpp_hip import ops
from .pointnet_pp_8dir import DIRS_8, BackboneBNHead, PointNetSetAbstraction  # noqa: F401


class PointNetPPFwd(BackboneBNHead):
    """forward(xyz (B,N,3)) -> (B,3) unit vector: F.normalize(fc3(x), dim=1) (models/pointnet_pp_Fwd.py:77-98)."""

    def __init__(self, sampler=None, grouper=None):
        super().__init__(sampler, grouper)
        self.fc3 = nn.Linear(256, 3)

    def forward(self, xyz, centres=None, drop_mask=None):
        return ops.l2_normalize(ops.fc_block(self.trunk(xyz, centres, drop_mask), self.fc3, training=self.training))
