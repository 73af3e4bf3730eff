"""models/base.py -- drop-in for the reference's point-set primitives (models/base.py:4-35).

Same names, argument meaning and return types (int64 indices); the work is done by the HIP
kernels in csrc/index_kernels.hip through libpnpp_hip.so.  GPU tensors only.
"""
import torch

from pnpp_hip import ops


def index_points(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """points (B,N,C), idx (B,S) or (B,S,K) -> rows of `points`, shape idx.shape + (C,).
    Reference: models/base.py:4-18 (differentiable w.r.t. `points`)."""
    return ops.index_points(points, idx)


def square_distance(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """src (B,N,3), dst (B,M,3) -> (B,N,M) squared distances, bit-equal to the reference's
    -2ab + a^2 + b^2 float32 evaluation (models/base.py:20-27)."""
    return ops.square_distance(src, dst)


def query_ball_point(new_xyz: torch.Tensor, xyz: torch.Tensor, nsample: int) -> torch.Tensor:
    """The reference's "ball query" is a kNN (models/base.py:29-35): indices (B,npoint,nsample),
    int64, of the nsample nearest points.  Order here is ascending (distance, index); the
    reference's topk(sorted=False) order is unspecified, compare as sets."""
    return ops.knn(new_xyz, xyz, nsample).long()
