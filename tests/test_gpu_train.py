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
    assert len(hist["train"]) == 6 and all(map(lambda v: v == v and v < 20.0, hist["train"]))   # finite, sane KL values
    sd = torch.load(tmp_path / "vonMises_best.pth")
    PointNetPPVonMises().load_state_dict(sd)                               # checkpoint uses the reference's keys
    assert test_kl == test_kl


def test_fixed_batch_is_overfitted():
    """Learning check that does not depend on 44 random clouds generalising in six epochs: 100 Adam steps on one fixed
    batch (fresh random centres and dropout every step) must pull the KL down by more than a nat."""
    from models.pointnet_pp_8dir import PointNetSetAbstraction
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from pnpp_hip import ops, optim, sampling
    import synthetic
    torch.manual_seed(42)
    sampling.reset(0)
    old = PointNetSetAbstraction.sampler
    PointNetSetAbstraction.sampler = "device"
    try:
        model = PointNetPPVonMises().cuda().train()
        opt = optim.FlatAdam(model.parameters(), lr=1e-3)
        xyz, mu, kap, _ = synthetic.rotated_clouds(16, 256, seed=5)
        xyz, mu, kap = xyz.cuda(), mu.cuda(), kap.cuda()
        hist = []
        for _ in range(100):
            opt.zero_grad()
            loss = ops.vm_head_kl_loss(model.features(xyz), mu, kap)
            loss.backward()
            opt.step()
            hist.append(loss.item())
    finally:
        PointNetSetAbstraction.sampler = old
    first, last = sum(hist[:10]) / 10, sum(hist[-10:]) / 10
    assert all(v == v for v in hist) and last < first - 1.0, (first, last)


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
