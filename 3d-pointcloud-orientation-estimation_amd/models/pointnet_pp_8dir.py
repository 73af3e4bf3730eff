"""models/pointnet_pp_8dir.py -- drop-in for the reference file of the same name:
PointNetSetAbstraction (the backbone block of every pointnet_pp_* model), DIRS_8, PointNetPP8Dir.

Parameter containers are the same nn.Conv2d / nn.BatchNorm2d / nn.Linear modules in the same
construction order, so state_dict keys, shapes and seeded default initialisation are identical to
the reference (models/pointnet_pp_8dir.py:6-19,58-74).  forward() runs on the HIP kernels.
"""
import os

import torch
import torch.nn as nn

from pnpp_hip import ops, sampling


class PointNetSetAbstraction(nn.Module):
    """Sample centres, group the nsample nearest points, 1x1 conv + BatchNorm + ReLU x len(mlp), max.

    Reference: models/pointnet_pp_8dir.py:6-43.  Two knobs the reference does not have:
      sampler  'randperm' (default) -- per cloud torch.randperm(N)[:npoint] on the CPU default
                                       generator, exactly the reference's draw (line 28);
               'device'             -- the same distribution drawn on the GPU (no host work);
               'fps'                -- true farthest point sampling (PointNet++Demo.py:8-29).
      grouper  'knn' (default, what the reference calls query_ball_point) or ('ball', radius)
               for the radius query of PointNet++Demo.py:49-70.
    Both are per-instance constructor keywords (every model class passes its own `sampler=` / `grouper=` down to its
    levels); the class attributes -- PNPP_SAMPLER / PNPP_GROUPER in the environment -- are only the default for
    instances built without them, so two models in one process can differ.
    """

    sampler = os.environ.get("PNPP_SAMPLER", "randperm")
    grouper = os.environ.get("PNPP_GROUPER", "knn")

    def __init__(self, npoint, nsample, in_channel, mlp_channels, group_all=False, sampler=None, grouper=None):
        super().__init__()
        self.npoint = npoint
        self.nsample = nsample
        self.group_all = group_all
        if sampler is not None:
            if sampler not in ("randperm", "device", "fps"):
                raise ValueError(f"unknown sampler '{sampler}'")
            self.sampler = sampler
        if grouper is not None:
            self.grouper = grouper

        last_ch = in_channel + 3
        self.convs = nn.ModuleList()
        self.bns = nn.ModuleList()
        for out_ch in mlp_channels:
            self.convs.append(nn.Conv2d(last_ch, out_ch, 1))
            self.bns.append(nn.BatchNorm2d(out_ch))
            last_ch = out_ch

    def _centres(self, xyz):
        B, N, _ = xyz.shape
        if self.sampler == "randperm":
            idx = torch.stack([torch.randperm(N)[:self.npoint] for _ in range(B)])
            return idx.to(xyz.device, non_blocking=True)
        if self.sampler == "device":
            return sampling.device_random_centres(B, N, self.npoint, xyz.device)
        if self.sampler == "fps":
            return ops.farthest_point_sample(xyz, self.npoint)
        raise ValueError(f"unknown sampler '{self.sampler}'")

    def forward(self, xyz, points, centre_idx=None, pre=None):
        """xyz (B,N,3), points (B,N,D) or None -> new_xyz (B,S,3), new_points (B,S,C_out).
        centre_idx (B,S) optionally injects the centres (tests, parity runs); pre = this level's workspace from
        ops.group_pair (centres gathered and neighbours found ahead of time, together with the level above / below)."""
        if self.group_all:
            return ops.set_abstraction(xyz, points, None, None, True, self.training, self.convs, self.bns)
        if pre is not None:
            return ops.set_abstraction(xyz, points, centre_idx, self.nsample, False, self.training, self.convs, self.bns, pre=pre)
        if centre_idx is None:
            centre_idx = self._centres(xyz)
        nbr = None
        if self.grouper != "knn":
            kind, radius = self.grouper if isinstance(self.grouper, tuple) else tuple(self.grouper.split(":"))
            if kind != "ball":
                raise ValueError(f"unknown grouper '{self.grouper}'")
            new_xyz = ops.index_points(xyz, centre_idx)
            nbr = ops.ball_query(float(radius), self.nsample, xyz, new_xyz)
        return ops.set_abstraction(xyz, points, centre_idx, self.nsample, False, self.training, self.convs, self.bns,
                                   neighbour_idx=nbr)


def stacked_levels(sa1, sa2, xyz, c1=None, c2=None):
    """sa1 on the cloud, then sa2 on sa1's centres (the first two lines of every pointnet_pp_* forward, e.g.
    models/pointnet_pp_vonMises.py:28-29) -> l1_xyz, l1_points, l2_xyz, l2_points.  When both levels group by kNN and both
    centre draws are known before sa1 runs (injected, or any sampler but 'fps', whose second draw needs sa1's centres), the
    two neighbour searches go out as ONE launch ahead of the MLPs (ops.group_pair); the draws happen in the reference's
    order (all of sa1's, then all of sa2's), so the CPU generator's sequence is unchanged."""
    pair_ok = (not sa1.group_all and not sa2.group_all and sa1.grouper == "knn" and sa2.grouper == "knn" and xyz.is_cuda
               and (c1 is not None or sa1.sampler != "fps") and (c2 is not None or sa2.sampler != "fps")
               and sa1.npoint <= xyz.shape[1] and sa2.npoint <= sa1.npoint and sa1.nsample <= xyz.shape[1] and sa2.nsample <= sa1.npoint)
    if not pair_ok:
        l1_xyz, l1_pts = sa1(xyz, None, c1)
        l2_xyz, l2_pts = sa2(l1_xyz, l1_pts, c2)
        return l1_xyz, l1_pts, l2_xyz, l2_pts
    if c1 is None:
        c1 = sa1._centres(xyz)
    if c2 is None:
        c2 = sa2._centres(xyz[:, :sa1.npoint])     # only the shape (B, npoint1, 3) enters a random draw
    pre1, pre2 = ops.group_pair(xyz, c1, c2, sa1.nsample, sa1.convs, sa2.nsample, sa2.convs, training=sa1.training)
    l1_xyz, l1_pts = sa1(xyz, None, c1, pre=pre1)
    l2_xyz, l2_pts = sa2(l1_xyz, l1_pts, c2, pre=pre2)
    return l1_xyz, l1_pts, l2_xyz, l2_pts


# the 8 horizontal directions (0, 45, ... 315 degrees), clockwise from the canonical forward axis [0,0,-1]
DIRS_8 = torch.tensor([
    [0.0000, 0.0, -1.0000],
    [0.7071, 0.0, -0.7071],
    [1.0000, 0.0, 0.0000],
    [0.7071, 0.0, 0.7071],
    [0.0000, 0.0, 1.0000],
    [-0.7071, 0.0, 0.7071],
    [-1.0000, 0.0, 0.0000],
    [-0.7071, 0.0, -0.7071],
])


class BackboneBNHead(nn.Module):
    """sa1 / sa2 / sa3 + fc1 / bn1 / fc2 / bn2 / drop: the part shared by PointNetPP8Dir, PointNetPPVonMises, PointNetPP,
    PointNetPPFwd, PointNetPPXYZ and PointNetPPXYZ_Schedmit (identical constructor lines in each of the reference's
    files, e.g. models/pointnet_pp_8dir.py:61-73).  Subclasses add their output layers AFTER super().__init__(), so
    parameter creation order -- and with it the seeded default initialisation -- matches the reference."""

    def __init__(self, sampler=None, grouper=None):
        super().__init__()
        kw = dict(sampler=sampler, grouper=grouper)
        self.sa1 = PointNetSetAbstraction(128, 32, 0, [64, 64, 128], **kw)
        self.sa2 = PointNetSetAbstraction(32, 32, 128, [128, 128, 256], **kw)
        self.sa3 = PointNetSetAbstraction(None, None, 256, [256, 512, 1024], group_all=True)

        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop = nn.Dropout(0.5)

    _presampled = None

    def use_presampled(self, ring) -> None:
        """ring: a pnpp_hip.sampling.CentreRing whose buffers hold the centres of the next forward pass (drawn one step ahead
        by the previous step's tail launch); None switches back to drawing at the start of the step."""
        self._presampled = ring

    def levels12(self, xyz, centres=None):
        """sa1 + sa2 -> (l2_xyz, l2_points); centres = (sa1 centre indices, sa2 centre indices) injects the draws."""
        B = xyz.size(0)
        if centres is None and self._presampled is not None and self.training and self._presampled.B == B:
            centres = self._presampled.centres
        c1, c2 = centres if centres is not None else (None, None)
        if (centres is None and self.sa1.sampler == "device" and self.sa2.sampler == "device" and not self.sa1.group_all
                and not self.sa2.group_all):
            # neither draw depends on data: both levels' centres from one launch (same centres as the per-level draws)
            c1, c2 = sampling.device_random_centres_pair(B, xyz.size(1), self.sa1.npoint, self.sa1.npoint, self.sa2.npoint, xyz.device)
        _, _, l2_xyz, l2_pts = stacked_levels(self.sa1, self.sa2, xyz, c1, c2)
        return l2_xyz, l2_pts

    def trunk(self, xyz, centres=None, drop_mask=None):
        """(B,N,3) -> (B,256): everything up to and including the dropout in front of the output layers.
        centres = (sa1 centre indices, sa2 centre indices) and drop_mask (B,256) inject the random draws (parity runs)."""
        B = xyz.size(0)
        l2_xyz, l2_pts = self.levels12(xyz, centres)
        _, l3_pts = self.sa3(l2_xyz, l2_pts)
        x = l3_pts.view(B, -1)
        x = ops.fc_block(x, self.fc1, self.bn1, relu=True, training=self.training)
        return ops.fc_block(x, self.fc2, self.bn2, relu=True, dropout=self.drop, training=self.training, mask=drop_mask)


class PointNetPP8Dir(BackboneBNHead):
    """PointNet++ backbone + 8-way direction head, raw logits out (models/pointnet_pp_8dir.py:58-85)."""

    def __init__(self, sampler=None, grouper=None):
        super().__init__(sampler, grouper)
        self.fc3 = nn.Linear(256, 8)

    def forward(self, xyz, centres=None, drop_mask=None):
        return ops.fc_block(self.trunk(xyz, centres, drop_mask), self.fc3, training=self.training)
