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
    # static batch shape + device-side sampler: every step but the first two of the run replays the captured hipGraph
    # (44 training clouds / batch 16 = 2 full + 1 ragged batch per epoch; the ragged one runs eagerly)
    assert hist["steps"]["graph"] >= 9 and hist["steps"]["eager"] <= 9, hist["steps"]


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
    # per-category rows carry numbers (the reference's format, train_multi_peaks_vonMises_KL.py:143-146), not nan
    row = [ln for ln in txt.splitlines() if ln.startswith("[synthetic]")][0]
    assert "nan" not in row and abs(float(row.split("Train=")[1].split()[0]) - hist["total"]["train"][-1]) < 1e-5, row
    assert hist["synthetic"]["val"] == hist["total"]["val"]                # one category: its curve is the total curve


def test_8dir_script(tmp_path, monkeypatch):
    import train_8dir_KL as t
    from models.pointnet_pp_8dir import PointNetPP8Dir
    hist, test = _run(t, tmp_path, monkeypatch, epochs=2)
    PointNetPP8Dir().load_state_dict(torch.load(tmp_path / "8dir_KLdiv_0926.pth"))     # the reference's file name (line 122)
    assert 0 < test < 10
    lines = (tmp_path / "summary.txt").read_text().splitlines()          # reference lines 147-149
    assert [ln.split("\t")[0] for ln in lines] == ["synthetic", "Overall"]
    assert abs(float(lines[1].split("\t")[1]) - test) < 1e-5 and abs(float(lines[0].split("\t")[1]) - test) < 1e-5


def test_device_clip_matches_torch_clip_grad_norm():
    """FlatAdam.clip_grad_norm_ (norm reduced on the device, the coefficient folded into the Adam launch -- no .item()) vs
    torch.nn.utils.clip_grad_norm_ + torch.optim.Adam (train_multi_peaks_vonMises_KL.py:235-236), for a gradient above
    and one below the threshold, and with the data-parallel convention (buffer = SUM over ranks, grad_scale = 1/world)."""
    from pnpp_hip import optim
    g = torch.Generator().manual_seed(0)
    for scale, world, zero in ((10.0, 1, False), (1e-3, 1, True), (10.0, 4, True)):
        shapes = [(64, 3), (64,), (128, 64), (7,)]
        ref = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
        mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
        o_ref = torch.optim.Adam(ref, lr=1e-3)
        o = optim.FlatAdam(mine, lr=1e-3)
        for it in range(3):
            grads = [scale * torch.randn(*s, generator=g).cuda() for s in shapes]     # the mean gradient
            for p, q, gr in zip(ref, mine, grads):
                p.grad = gr.clone()
                q.grad.copy_(gr * world)                                              # what the all-reduce leaves: the sum
            n_ref = torch.nn.utils.clip_grad_norm_(ref, max_norm=1.0)
            n_mine = o.clip_grad_norm_(1.0)                                           # device tensor, buffer's own norm
            assert abs(float(n_mine) / world - float(n_ref)) <= 1e-5 * float(n_ref)
            o_ref.step()
            o.step(grad_scale=1.0 / world, zero_grad=zero)
            assert (float(o.flat_g.abs().max()) == 0.0) == zero
            for p, q in zip(ref, mine):
                assert torch.allclose(p, q, rtol=0, atol=2e-6), (scale, world, it)
        # the clip is consumed by exactly one step: the next one is unclipped again
        assert o._pending_clip is None


def test_trainer_graph_path_equals_eager_path(tmp_path):
    """trainer.fit on a static-shape loader: the hipGraph path (what the scripts now run with --sampler device) and the
    eager path produce the same per-epoch losses and the same final weights, bit for bit (same kernels, same order)."""
    import copy
    from models.pointnet_pp_mvM import PointNetPPMvM
    from pnpp_hip import ops, sampling, trainer
    import synthetic
    xyz, _, _, fwd = synthetic.rotated_clouds(48, 256, seed=11)
    K = torch.tensor([1, 2, 4])[torch.randint(0, 3, (48,), generator=torch.Generator().manual_seed(1))]
    data = [xyz, synthetic.multi_peak_gt(fwd, K), K]
    dev = torch.device("cuda")

    def loss(model, batch):
        return ops.match_loss(*model(batch[0]), batch[1], batch[2])

    out = {}
    for mode in (True, False):
        torch.manual_seed(3)
        sampling.reset(0)
        model = PointNetPPMvM(sampler="device", p_drop=0.0).to(dev)
        loaders = {"train": trainer.SyntheticLoader(data, 16, False, dev), "val": trainer.SyntheticLoader(data, 16, False, dev)}
        hist, best, _ = trainer.fit(model, loss, loaders, 3, 1e-3, dev, clip_norm=1.0, log=lambda *_: None, use_graph=mode,
                                    label_index=3, n_labels=1)
        out[mode] = (hist, copy.deepcopy(model.state_dict()))
    hg, he = out[True][0], out[False][0]
    assert hg["steps"] == {"graph": 8, "eager": 1} and he["steps"] == {"graph": 0, "eager": 9}
    assert hg["train"] == he["train"] and hg["val"] == he["val"], (hg["train"], he["train"])
    assert hg["labels"][0]["train"] == hg["train"]
    for k, v in out[True][1].items():
        assert torch.equal(v, out[False][1][k]), k


def test_gradient_accumulation_with_fused_sinks():
    """FlatAdam(fused_grads=True): the first backward pass of a window overwrites each parameter's slice of the flat
    buffer, a second pass before step()/zero_grad() must ADD to it (micro-batch accumulation, two losses, shared
    weights) instead of silently replacing it."""
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from pnpp_hip import ops, optim
    import synthetic
    torch.manual_seed(42)
    model = PointNetPPVonMises(sampler="device").cuda().train()
    model.drop.p = 0.0
    opt = optim.FlatAdam(model.parameters(), lr=1e-3)
    xyz, mu, kap, _ = synthetic.rotated_clouds(8, 256, seed=5)
    xyz, mu, kap = xyz.cuda(), mu.cuda(), kap.cuda()
    g = torch.Generator().manual_seed(0)
    c = [[torch.stack([torch.randperm(n, generator=g)[:s] for _ in range(4)]).cuda() for n, s in ((256, 128), (128, 32))]
         for _ in range(2)]

    def half(i):   # BatchNorm statistics are per call: the two halves are two independent micro-batches
        sl = slice(4 * i, 4 * i + 4)
        return ops.vm_head_kl_loss(model.features(xyz[sl], centres=c[i]), mu[sl], kap[sl])

    singles = []
    for i in range(2):
        opt.zero_grad()
        half(i).backward()
        singles.append(opt.flat_g.clone())
    opt.zero_grad()
    half(0).backward()
    half(1).backward()                                    # no zero_grad in between: accumulate
    want = singles[0] + singles[1]
    assert float((opt.flat_g - want).abs().max()) <= 1e-6 * float(want.abs().max())
    assert float(singles[1].abs().max()) > 0 and not torch.equal(opt.flat_g, singles[1])
    opt.zero_grad()
    (half(0) + half(1)).backward()                        # two passes summed into one loss
    assert float((opt.flat_g - want).abs().max()) <= 1e-6 * float(want.abs().max())
