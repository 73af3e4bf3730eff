"""Drop-in `models` package.  The reference's models/__init__.py:1-9 re-exports seven classes, five of
which (vanilla PointNet, PointTransformer, the xyz / Schmidt / Fwd heads) are outside the hot path this
repository implements; the set-abstraction models are exported here
and the training scripts import the submodules directly, as the reference's scripts do."""
from .pointnet_pp_8dir import PointNetPP8Dir, PointNetSetAbstraction, DIRS_8
from .pointnet_pp_vonMises import PointNetPPVonMises
from .pointnet_pp_mvM import PointNetPPMvM

__all__ = ["PointNetPP8Dir", "PointNetSetAbstraction", "DIRS_8", "PointNetPPVonMises", "PointNetPPMvM"]
