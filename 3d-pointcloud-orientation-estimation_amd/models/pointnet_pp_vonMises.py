"""models/pointnet_pp_vonMises.py -- drop-in for the reference's single-peak von-Mises model
(models/pointnet_pp_vonMises.py:8-38): PointNet++ backbone + (mu, kappa) head."""
import torch.nn as nn

from pnpp_hip import ops
from .pointnet_pp_8dir import BackboneBNHead, PointNetSetAbstraction  # noqa: F401


class PointNetPPVonMises(BackboneBNHead):
    """forward(xyz (B,N,3)) -> (mu (B,) in [-pi, pi], kappa (B,) >= 0)."""

    def __init__(self, sampler=None, grouper=None):
        super().__init__(sampler, grouper)
        self.fc3 = nn.Linear(256, 2)

    def features(self, xyz, centres=None, drop_mask=None):
        """The raw fc3 output (B,2), in front of the tanh / softplus activations (and of the fused loss kernels)."""
        return ops.fc_block(self.trunk(xyz, centres, drop_mask), self.fc3, training=self.training)

    def forward(self, xyz, centres=None, drop_mask=None):
        return ops.vm_head(self.features(xyz, centres, drop_mask))       # mu = tanh(out[:,0]) * pi, kappa = softplus(out[:,1])
