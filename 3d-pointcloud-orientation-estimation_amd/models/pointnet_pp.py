"""models/pointnet_pp.py -- drop-in for the reference file of the same name (models/pointnet_pp.py:6-68):
PointNetSetAbstraction (the same block as in pointnet_pp_8dir) and PointNetPP, the backbone with a raw 3-vector head."""
import torch.nn as nn

from pnpp_hip import ops
from .pointnet_pp_8dir import PointNetSetAbstraction  # noqa: F401  (re-exported: the reference defines it per file)


class _BackboneBNHead(nn.Module):
    """sa1/sa2/sa3 + fc1/bn1/fc2/bn2/drop shared by PointNetPP, PointNetPPFwd, PointNetPPXYZ and PointNetPPXYZ_Schedmit
    (models/pointnet_pp.py:46-56 and the identical constructors of the other three files).  Subclasses add their
    output layers AFTER calling super().__init__() so that parameter creation order -- and with it the seeded default
    initialisation -- matches the reference."""

    def __init__(self):
        super().__init__()
        self.sa1 = PointNetSetAbstraction(128, 32, 0, [64, 64, 128])
        self.sa2 = PointNetSetAbstraction(32, 32, 128, [128, 128, 256])
        self.sa3 = PointNetSetAbstraction(None, None, 256, [256, 512, 1024], group_all=True)

        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop = nn.Dropout(0.5)

    def features(self, xyz, centres=None, drop_mask=None):
        """(B,N,3) -> (B,256): everything up to and including the dropout in front of the output layers."""
        B = xyz.size(0)
        c1, c2 = centres if centres is not None else (None, None)
        l1_xyz, l1_pts = self.sa1(xyz, None, c1)
        l2_xyz, l2_pts = self.sa2(l1_xyz, l1_pts, c2)
        _, l3_pts = self.sa3(l2_xyz, l2_pts)
        x = l3_pts.view(B, -1)
        x = ops.fc_block(x, self.fc1, self.bn1, relu=True, training=self.training)
        return ops.fc_block(x, self.fc2, self.bn2, relu=True, dropout=self.drop, training=self.training, mask=drop_mask)


class PointNetPP(_BackboneBNHead):
    """forward(x (B,N,3)) -> (B,3) raw direction vector (models/pointnet_pp.py:44-68)."""

    def __init__(self):
        super().__init__()
        self.fc3 = nn.Linear(256, 3)

    def forward(self, x, centres=None, drop_mask=None):
        return ops.fc_block(self.features(x, centres, drop_mask), self.fc3, training=self.training)
