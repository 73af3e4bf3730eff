"""GPU: the drop-in training scripts run end to end on generated clouds (no dataset ships with the reference):
a few epochs through the same loop a user of the reference would run, loss goes down, checkpoints load."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(mod, tmp_path, monkeypatch, n=64, epochs=3, batch=16, points=256):
    monkeypatch.setattr(mod, "RES", tmp_path)
    if hasattr(mod, "FIGS"):
        monkeypatch.setattr(mod, "FIGS", tmp_path / "figs")
    monkeypatch.setattr(mod, "EPOCHS", epochs)
    monkeypatch.setattr(mod, "BATCH", batch)
    monkeypatch.setattr(mod, "NUM_POINTS", points)
    return mod.main(["--synthetic", str(n), "--sampler", "device"])


def test_single_peak_script(tmp_path, monkeypatch):
    import train_single_peak_vonMises_KL as t
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    hist, test_kl = _run(t, tmp_path, monkeypatch, epochs=6)
    assert len(hist["train"]) == 6 and all(map(lambda v: v == v, hist["train"]))
    assert min(hist["train"][3:]) < hist["train"][0]                      # it learns something on 44 clouds
    sd = torch.load(tmp_path / "vonMises_best.pth")
    PointNetPPVonMises().load_state_dict(sd)                               # checkpoint uses the reference's keys
    assert test_kl == test_kl


def test_multi_peak_script(tmp_path, monkeypatch):
    import train_multi_peaks_vonMises_KL as t
    from models.pointnet_pp_mvM import PointNetPPMvM
    hist, test_kl = _run(t, tmp_path, monkeypatch, epochs=2)
    PointNetPPMvM().load_state_dict(torch.load(tmp_path / "mvM_best.pth"))
    txt = (tmp_path / "results.txt").read_text()
    assert txt.startswith("=== Multi-Peak von Mises KL Summary ===") and "Test KL:" in txt


def test_8dir_script(tmp_path, monkeypatch):
    import train_8dir_KL as t
    from models.pointnet_pp_8dir import PointNetPP8Dir
    hist, test = _run(t, tmp_path, monkeypatch, epochs=2)
    PointNetPP8Dir().load_state_dict(torch.load(tmp_path / "8dir_best.pth"))
    assert 0 < test < 10
