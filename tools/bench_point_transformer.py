#!/usr/bin/env python3
"""Timing of one training step of BASELINE config 5 (models/point_transformer.py, N=4096, 8 clouds per GPU = batch 64
over 8 GPUs) on one MI355X: forward + MSE harness loss + backward + fused Adam (--dropout sets the encoder layers' dropout probability).  Not the driver's bench line (that is bench.py / configs[1]); prints one JSON line
in bench.py's format (tools/benchline.py: `roofline` for its costliest kernel) plus the per-kernel time table and the attention kernels' TFLOP/s."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-pointcloud-orientation-estimation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dropout", type=float, default=0.0, help="dropout probability of the encoder layers (reference default 0.1)")
    a = ap.parse_args()
    from models.point_transformer import PointTransformer
    from pnpp_hip import _lib, ops, optim
    import synthetic
    torch.manual_seed(42)
    model = PointTransformer().cuda().train().set_dropout(a.dropout)
    opt = optim.FlatAdam(model.parameters(), lr=1e-3)
    xyz, _, _, fwd = synthetic.rotated_clouds(a.batch, a.points, seed=1234)
    xyz, tgt = xyz.cuda(), fwd.cuda()

    def step():
        opt.zero_grad()
        loss = ops.mse_loss(model(xyz), tgt)
        loss.backward()
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    import benchline
    rows = benchline.profiled_rows(step, 1)
    H, dh, L = 4, 16, len(model.transformer.layers)
    att_flops_fwd = 4.0 * a.batch * H * a.points * a.points * dh          # QK^T and PV, 2 flops per MAC
    table = []
    for tag, cnt, ms in rows[:12]:
        e = {"kernel": tag, "launches": cnt, "ms": round(ms, 3)}
        if tag.startswith("attention_fwd"):
            e["tflops"] = round(att_flops_fwd * cnt / (ms * 1e-3) / 1e12, 1)
        if tag.startswith("attention_bwd_dq"):
            e["tflops"] = round(1.5 * att_flops_fwd * cnt / (ms * 1e-3) / 1e12, 1)   # S, dP, dQ
        if tag.startswith("attention_bwd_dkv"):
            e["tflops"] = round(2.0 * att_flops_fwd * cnt / (ms * 1e-3) / 1e12, 1)   # S, dP, dV, dK
        table.append(e)
    print(json.dumps(benchline.line(f"configs[4]: models/point_transformer.py N={a.points} batch={a.batch}/GPU, fwd+MSE+bwd+Adam, "
                                    f"dropout p={a.dropout}", a.batch, dt, a.steps, a.warmup, rows, 1, layers=L,
                                    final_loss=float(loss.detach()), top_kernels=table)))


if __name__ == "__main__":
    main()
