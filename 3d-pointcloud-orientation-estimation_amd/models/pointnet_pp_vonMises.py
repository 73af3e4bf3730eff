"""models/pointnet_pp_vonMises.py -- drop-in for the reference's single-peak von-Mises model
(models/pointnet_pp_vonMises.py:8-38): PointNet++ backbone + (mu, kappa) head."""
import torch
import torch.nn as nn

from pnpp_hip import ops
from .pointnet_pp_8dir import PointNetSetAbstraction


class PointNetPPVonMises(nn.Module):
    """forward(xyz (B,N,3)) -> (mu (B,) in [-pi, pi], kappa (B,) >= 0)."""

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
        self.fc3 = nn.Linear(256, 2)

    def features(self, xyz, centres=None, drop_mask=None):
        """Everything up to the raw fc3 output (B,2)."""
        B = xyz.size(0)
        c1, c2 = centres if centres is not None else (None, None)
        l1_xyz, l1_pts = self.sa1(xyz, None, c1)
        l2_xyz, l2_pts = self.sa2(l1_xyz, l1_pts, c2)
        _, l3_pts = self.sa3(l2_xyz, l2_pts)
        x = l3_pts.view(B, -1)
        x = ops.fc_block(x, self.fc1, self.bn1, relu=True, training=self.training)
        x = ops.fc_block(x, self.fc2, self.bn2, relu=True, dropout=self.drop, training=self.training, mask=drop_mask)
        return ops.fc_block(x, self.fc3, training=self.training)

    def forward(self, xyz, centres=None, drop_mask=None):
        out = self.features(xyz, centres, drop_mask)
        return ops.vm_head(out)       # mu = tanh(out[:,0]) * pi, kappa = softplus(out[:,1])
