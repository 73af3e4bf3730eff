"""Drop-in `models` package.  The reference's models/__init__.py:1-9 re-exports seven classes; the five built on the
set-abstraction backbone are here (plus the von-Mises models the training scripts import from their submodules).
Vanilla PointNet and PointTransformer are outside the hot path this repository implements."""
from .pointnet_pp import PointNetPP
from .Pointnet_pp_xyz import PointNetPPXYZ
from .Pointnet_pp_xyz_Schedmit import PointNetPPXYZ_Schedmit
from .pointnet_pp_8dir import PointNetPP8Dir, PointNetSetAbstraction, DIRS_8
from .pointnet_pp_Fwd import PointNetPPFwd
from .pointnet_pp_vonMises import PointNetPPVonMises
from .pointnet_pp_mvM import PointNetPPMvM

__all__ = ["PointNetPP", "PointNetPPXYZ", "PointNetPPXYZ_Schedmit", "PointNetPP8Dir", "PointNetPPFwd",
           "PointNetSetAbstraction", "DIRS_8", "PointNetPPVonMises", "PointNetPPMvM"]
