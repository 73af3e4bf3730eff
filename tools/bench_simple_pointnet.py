#!/usr/bin/env python3
"""Times one training step of the drop-in SimplePointNet (BASELINE configs[0]; simple_pointnet_train.py) on the GPU at the
configuration BASELINE.json quotes (256 points, batch 4) and at the script's own defaults (10,000 points, batch 16), and
prints one line per size in bench.py's format (tools/benchline.py: `roofline` for the costliest kernel) and the per-kernel event
table of the larger one.  Not the driver's bench line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-pointcloud-orientation-estimation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

import simple_pointnet_train as spt  # noqa: E402
import synthetic  # noqa: E402
from pnpp_hip import optim  # noqa: E402


def run(B, N, steps=50, warmup=10, profile=False):
    torch.manual_seed(42)
    model = spt.SimplePointNet().cuda().train()
    opt = optim.FlatAdam(model.parameters(), lr=1e-3)
    xyz, _, _, fwd = synthetic.rotated_clouds(B, N, seed=1)
    xyz, fwd = xyz.cuda(), fwd.float().cuda()

    def step():
        opt.zero_grad()
        spt.criterion(model(xyz), fwd).mean().backward()
        opt.step()

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    import benchline
    rows = benchline.profiled_rows(step, 5)
    rec = benchline.line(f"configs[0]: simple_pointnet_train.py SimplePointNet fwd+MSE+bwd+Adam, eager launches, B={B} N={N}", B, ms * 1e-3,
                         steps, warmup, rows, 5, points_per_s=B * N / ms * 1e3)
    if profile:
        rec["kernel_table"] = "\n".join(f"{1e3 * t_ms / 5:9.1f} us/step  {c / 5:4.1f} x {1e3 * t_ms / c:8.1f} us  {t}" for t, c, t_ms in rows)
    return rec


if __name__ == "__main__":
    out = [run(4, 256), run(16, 10_000, profile=True)]
    for r in out:
        tab = r.pop("kernel_table", None)
        print(json.dumps(r))
        if tab:
            print(tab)
