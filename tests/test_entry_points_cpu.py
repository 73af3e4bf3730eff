"""CPU tests of the drop-in entry points: dataloader_* (reference tuple layouts, error behaviour) and the
module-level surface of the train_* scripts.  The reference's own behaviour for each case is cited inline."""
import os

import numpy as np
import pytest
import torch

PLY_HEADER = "ply\nformat ascii 1.0\nelement vertex {n}\nproperty float x\nproperty float y\nproperty float z\nend_header\n"


def _write_ply(path, pts, extra_cols=0):
    with open(path, "w") as f:
        f.write(PLY_HEADER.format(n=len(pts)))
        for p in pts:
            f.write(" ".join(f"{v:.6f}" for v in list(p) + [0.5] * extra_cols) + "\n")


@pytest.fixture()
def cloud(tmp_path):
    pts = np.random.RandomState(0).rand(50, 3).astype(np.float32)
    p = tmp_path / "chair_0001.ply"
    _write_ply(p, pts, extra_cols=3)        # normals after xyz are ignored (first three columns only)
    return p, pts


def test_read_ply_and_sampling(cloud, tmp_path, monkeypatch):
    import dataloader_common as dc
    p, pts = cloud
    got = dc.read_ply(p)
    assert got.shape == (50, 3) and np.allclose(got, pts, atol=1e-6)
    np.random.seed(0)
    s = dc.sample_pts(got, 20)
    assert s.shape == (20, 3) and len({tuple(r) for r in s.tolist()}) == 20          # without replacement when enough points
    s = dc.sample_pts(got, 200)
    assert s.shape == (200, 3)                                                        # with replacement otherwise
    monkeypatch.setenv("PNPP_PLY_CACHE_DIR", str(tmp_path / "cache"))
    a = dc.read_ply(p)
    b = dc.read_ply(p)                                                                # second read is served by the cache
    assert np.array_equal(np.asarray(a), np.asarray(b)) and len(os.listdir(tmp_path / "cache")) == 1
    # a damaged cache file (a writer that died, a foreign file) is a miss: the source is parsed again and the cache replaced
    cfile = tmp_path / "cache" / os.listdir(tmp_path / "cache")[0]
    cfile.write_bytes(b"\x93NUMPY garbage")
    os.utime(cfile, (os.path.getmtime(p) + 10, os.path.getmtime(p) + 10))
    c = dc.read_ply(p)
    assert np.array_equal(np.asarray(c), np.asarray(a))
    assert np.array_equal(np.load(cfile), np.asarray(a)) and os.listdir(tmp_path / "cache") == [cfile.name]   # no temporaries left


def test_single_peak_dataset(cloud):
    from dataloader_single_peak_vonMises import PointCloudDatasetVonMises, read_ply, sample_pts  # noqa: F401
    p, _ = cloud
    gt = p.with_name(p.stem + "_single_peak_vM_gt.txt")
    gt.write_text("# mu(rad)\tkappa\n-0.84729610\t8.000000\n")
    ds = PointCloudDatasetVonMises([(p, "chair")], 64)
    xyz, vm, lbl = ds[0]
    assert xyz.shape == (64, 3) and xyz.dtype == torch.float32 and vm.dtype == torch.float32 and lbl == 0
    assert torch.allclose(vm, torch.tensor([-0.84729610, 8.0]))
    gt.write_text("# nothing parsable\nfoo bar\n")                                     # reference: silently (0, 0)
    assert torch.equal(ds[0][1], torch.zeros(2))
    gt.write_text("0.5 -3.0\n")                                                        # kappa clamped at 0
    assert torch.allclose(ds[0][1], torch.tensor([0.5, 0.0]))
    gt.unlink()
    assert torch.equal(ds[0][1], torch.zeros(2))                                        # missing file too
    ds2 = PointCloudDatasetVonMises([(p, "sofa"), (p, "chair")], 8, label_map={"chair": 5, "sofa": 7})
    assert ds2[0][2] == 7 and len(ds2) == 2


def test_multi_peak_dataset(cloud, tmp_path):
    from dataloader_multi_peak_vonMises import PointCloudDatasetMvM
    p, _ = cloud
    gt = tmp_path / "chair_0001_multi_peak_vM_gt.txt"
    gt.write_text("K 2\nmu kappa weight\n0.100 8.0 0.5\n-3.0415 8.0 0.5\n")
    ds = PointCloudDatasetMvM([(str(p), str(gt), "chair")], 32, max_K=4)
    xyz, vm, K, lbl = ds[0]
    assert xyz.shape == (32, 3) and vm.shape == (4, 3) and K == 2 and lbl.dtype == torch.long and int(lbl) == 0
    assert torch.allclose(vm[:2], torch.tensor([[0.1, 8.0, 0.5], [-3.0415, 8.0, 0.5]])) and torch.all(vm[2:] == 0)
    gt.write_text("K 2\n")
    with pytest.raises(RuntimeError, match="too short"):
        ds[0]
    gt.write_text("K\nheader\n")
    with pytest.raises(RuntimeError, match="K line"):
        ds[0]
    with pytest.raises(FileNotFoundError):
        PointCloudDatasetMvM([(str(p) + ".missing", str(gt), "chair")], 32)[0]
    with pytest.raises(FileNotFoundError):
        PointCloudDatasetMvM([(str(p), str(gt) + ".missing", "chair")], 32)[0]


def test_8dir_dataset(cloud, tmp_path):
    from dataloader_8dir_sampled import PointCloudDataset
    p, _ = cloud
    prob = tmp_path / "chair_0001_8dir.txt"
    vals = np.array([0.5, 0.25, 0, 0, 0, 0, 0, 0.25], np.float32)
    np.savetxt(prob, vals)
    ds = PointCloudDataset([(p, prob, "chair"), (p, prob, "bottle"), (p, str(prob) + ".missing", "chair")], 16, {"bottle"})
    xyz, pr, lbl = ds[0]
    assert xyz.shape == (16, 3) and torch.allclose(pr, torch.from_numpy(vals)) and lbl == 0
    assert torch.allclose(ds[1][1], torch.full((8,), 0.125)) and ds[1][2] == 1       # symmetric class -> uniform
    assert torch.allclose(ds[2][1], torch.full((8,), 0.125))                           # missing file -> uniform
    prob.write_text("not numbers\n")
    assert torch.allclose(ds[0][1], torch.full((8,), 0.125))                           # unreadable -> uniform


def test_train_scripts_surface(monkeypatch):
    """Names the reference scripts define at module level; importing must not start training."""
    import train_single_peak_vonMises_KL as t1
    import train_multi_peaks_vonMises_KL as t2
    import train_8dir_KL as t3
    for name in ("ROOT", "RES", "FIGS", "NUM_POINTS", "BATCH", "EPOCHS", "LR", "SEED", "device", "kl_von_mises", "plot_curve"):
        assert hasattr(t1, name), name
    assert (t1.NUM_POINTS, t1.BATCH, t1.EPOCHS, t1.LR, t1.SEED) == (10_000, 16, 200, 1e-3, 42)       # reference lines 18-19
    for name in ("ROOT", "PLY_ROOT", "RES", "kl_von_mises", "match_loss", "write_summary_txt", "main"):
        assert hasattr(t2, name), name
    assert (t2.NUM_POINTS, t2.BATCH, t2.EPOCHS) == (10_000, 16, 100)
    assert hasattr(t3, "kl_loss_per_sample_from_logits")
    # multi-peak KL helper vs the first block of the reference's debug log (kappa_g = 0 -> clamp to 1e-6)
    v = t2.kl_von_mises(torch.tensor([0.0]), torch.tensor([1.2209635]), torch.tensor([-0.5524741]), torch.tensor([0.0]))
    assert abs(float(v) - 0.29115965962409973) < 2e-6


def test_simple_pointnet_script_surface(cloud, tmp_path):
    """BASELINE configs[0]: the names simple_pointnet_train.py defines, its dataset's tuple layout and errors
    (reference :46-81), the module's state_dict (reference :87-101), and the refusal to run without the GPU."""
    import simple_pointnet_train as spt
    for name in ("read_ply", "sample_points", "PointCloudDataset", "SimplePointNet", "train_model", "test_model", "main"):
        assert hasattr(spt, name), name
    assert (spt.NUM_POINTS, spt.BATCH, spt.EPOCHS, spt.LR, spt.SEED) == (10_000, 16, 200, 1e-3, 42)   # :226,231,246,244,197
    p, pts = cloud
    p.with_suffix(".txt").write_text("0.5 0.0 -0.8660254 ignored\n")
    ds = spt.PointCloudDataset(str(p.parent), [p.name], num_points=64)
    xyz, target = ds[0]
    assert len(ds) == 1 and xyz.shape == (64, 3) and xyz.dtype == torch.float32
    assert torch.allclose(target, torch.tensor([0.5, 0.0, -0.8660254]))
    p.with_suffix(".txt").write_text("0.5 0.0\n")
    with pytest.raises(ValueError):
        ds[0]
    p.with_suffix(".txt").unlink()
    with pytest.raises(FileNotFoundError):
        ds[0]
    with pytest.raises(RuntimeError):
        spt.read_ply(str(tmp_path / "missing.ply"))
    sd = spt.SimplePointNet().state_dict()
    shapes = {"conv1.weight": (64, 3, 1), "conv2.weight": (128, 64, 1), "conv3.weight": (256, 128, 1), "bn3.running_var": (256,),
              "fc1.weight": (128, 256), "bn4.weight": (128,), "fc2.weight": (3, 128), "fc2.bias": (3,)}
    for k, shp in shapes.items():
        assert tuple(sd[k].shape) == shp, k
    assert sum(v.numel() for k, v in sd.items() if "running" not in k and "tracked" not in k) == 76_035
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no GPU"):
            spt.main(["--synthetic", "8", "--points", "256", "--batch", "4", "--epochs", "1"])


def test_results_txt_format(tmp_path):
    import train_multi_peaks_vonMises_KL as t2
    hist = {"total": {"train": [0.2, 0.084364], "val": [0.3, 0.083457]}, "bathtub": {"train": [0.7, 0.71], "val": [0.8, 0.77]},
            "empty": {"train": [], "val": []}}
    out = tmp_path / "results.txt"
    t2.write_summary_txt(out, ["bathtub", "empty"], hist, test_kl=0.077724, best_val_epoch=55)
    lines = out.read_text().splitlines()
    # results/multi_peak_vonMises_KL/results.txt:1-6 of the reference
    assert lines[0] == "=== Multi-Peak von Mises KL Summary ===" and lines[1] == "Best Total Val Epoch: 55"
    assert lines[2] == "Test KL: 0.077724" and lines[4] == "-- Per-Category (last epoch) --"
    assert lines[5] == "[TOTAL] Train=0.084364 Val=0.083457" and lines[6] == "[bathtub] Train=0.710000 Val=0.770000"
    assert lines[7] == "[empty] Train=nan Val=nan"


def test_committed_bench_line_follows_the_contract():
    """The bench line committed under profiles/ (the one the roofline table in DESIGN.md quotes) carries every field of
    the driver's contract, the roofline and cpu_baseline objects included."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rounds = sorted(f for f in os.listdir(os.path.join(root, "profiles")) if f.startswith("round") and f.endswith("_bench.json"))
    assert "round1_final_bench.json" in rounds and len(rounds) >= 2        # one committed line per round
    for name in rounds:
        _check_bench_line(json.loads(open(os.path.join(root, "profiles", name)).read().strip().splitlines()[-1]))


def _check_bench_line(d):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "clouds/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] is not None
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert abs(d["value"] - d["n_gpus"] * d["config"]["per_gpu_batch"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_every_launch_that_can_lead_the_step_has_a_cost_model():
    """bench.py reports as dominant the largest launch it can price: every launch of the committed per-launch table above 10 us must
    have a cost model (round 2's review: a kernel without one was skipped silently), and the models must be positive."""
    import importlib.util, os, re, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_costs", os.path.join(root, "bench.py"))
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    seen = 0
    tables = sorted(f for f in os.listdir(os.path.join(root, "profiles")) if re.fullmatch(r"round\d+_kernel_table_events\.txt", f))
    for line in open(os.path.join(root, "profiles", tables[-1])):   # the latest round's per-launch table
        m = re.match(r"\s*([\d.]+) us/step\s+([\d.]+) x\s+([\d.]+) us\s+(.*)$", line)
        if not m or float(m.group(3)) < 10.0:
            continue
        tag = m.group(4).strip()
        if tag.startswith("vm_fc_head_kl_step_kernel"):   # one workgroup of float64 chains: latency, neither roof
            continue
        cost = bench.kernel_cost(tag)
        assert cost is not None, tag
        assert cost[0] >= 0.0 and cost[1] > 0.0, (tag, cost)
        seen += 1
    assert seen >= 15
