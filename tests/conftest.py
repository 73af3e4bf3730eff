"""pytest configuration.

Markers
  gpu : needs a real MI355X (run by the driver with `-m gpu` on the GPU box); everything
        else must pass on a CPU-only container (`-m "not gpu"`).

The drop-in package directory (`3d-pointcloud-orientation-estimation_amd/`) plays the role of
the reference's repository root: it is put on sys.path so that `models.*`, `dataloader_*`
and `train_*` import exactly as they do in the reference.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3d-pointcloud-orientation-estimation_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs an AMD MI355X GPU (HIP extension is exercised)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import restatement
    restatement.build_c_oracle()
    return restatement


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
