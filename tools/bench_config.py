#!/usr/bin/env python3
"""Bench lines for BASELINE.json configs[2] and configs[3] on ONE GPU, with the fields of the headline line (bench.py):

    python tools/bench_config.py --config 2   # models/pointnet_pp_mvM.py + match_loss, N=1024, batch 32
                                              # (train_multi_peaks_vonMises_KL.py:212-237: forward, matched KL, backward,
                                              #  clip_grad_norm_(1.0), Adam)
    python tools/bench_config.py --config 3   # models/pointnet_pp_8dir.py + soft-label cross entropy, N=2048, 32 clouds per GPU
                                              # (train_8dir_KL.py:85-97; BASELINE's global batch 256 = 8 GPUs x 32)

One step = zero_grad + forward (device-side centre sampling) + loss + backward replayed from one hipGraph, then the fused Adam
(config 2: with the device-side clip coefficient).  Prints one JSON line: metric / value / ms_per_step / roofline (dominant kernel
from dispatch-attached HIP events in a separate eager pass, priced by bench.kernel_cost) / mfma_fraction / hbm_fraction /
top_kernels.  Not the driver's bench line; committed under profiles/ once per round."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3d-pointcloud-orientation-estimation_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import bench  # noqa: E402
import synthetic  # noqa: E402
from pnpp_hip import _lib, ops, optim  # noqa: E402
from pnpp_hip.graph import GraphedStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, choices=[2, 3], required=True)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    _lib.lib()
    dev = torch.device("cuda:0")
    B = args.batch
    torch.manual_seed(42)
    if args.config == 2:
        from models.pointnet_pp_mvM import PointNetPPMvM
        N = 1024
        model = PointNetPPMvM(sampler="device").to(dev).train()
        xyz, _, _, fwd = synthetic.rotated_clouds(B, N, seed=1234)
        K = torch.tensor([(1, 2, 4)[i % 3] for i in range(B)])
        vm_gt = synthetic.multi_peak_gt(fwd, K).to(dev)
        K = K.to(dev)
        xyz = xyz.to(dev)
        inputs = [xyz, vm_gt, K]

        # the step's tail -- three output heads, head activations, match_loss, mean, their backward -- is ONE launch, which also
        # carries the next step's centre draw (PNPP_FUSED_TAIL=0: the separate launches; same arithmetic)
        fused_tail = os.environ.get("PNPP_FUSED_TAIL", "1") != "0"
        ring = None
        if fused_tail:
            from pnpp_hip import sampling
            ring = sampling.CentreRing(B, N, model.sa1.npoint, model.sa2.npoint, dev)
            model.use_presampled(ring)

        def loss_fn(x, g, k):
            if fused_tail:
                return model.loss_backward(x, g, k, next_centres=ring.job())
            mu, kappa, w = model(x)
            return ops.match_loss(mu, kappa, w, g, k).mean()
        clip = 1.0
        what = ("configs[2]: models/pointnet_pp_mvM.py multi-peak von-Mises KL (match_loss, K in {1,2,4}), N=1024, batch=32, "
                "fwd+loss+bwd+clip_grad_norm_(1.0)+Adam, random-init weights (seed 42), device-side centre sampling")
        metric = "clouds/sec fwd+bwd, pointnet_pp_mvM + match_loss N=1024"
        flops_per_cloud = bench.FLOPS_PER_CLOUD     # same backbone; the LayerNorm head is 0.1 % of it
        bytes_per_cloud = bench.BYTES_PER_CLOUD
    else:
        from models.pointnet_pp_8dir import PointNetPP8Dir, DIRS_8
        N = 2048
        model = PointNetPP8Dir(sampler="device").to(dev).train()
        xyz, _, _, fwd = synthetic.rotated_clouds(B, N, seed=1234)
        prob = synthetic.dir8_soft_labels(fwd.float(), DIRS_8).to(dev)
        xyz = xyz.to(dev)
        inputs = [xyz, prob]

        def loss_fn(x, p):
            return ops.soft_ce(model(x), p).mean()
        clip = None
        what = ("configs[3]: models/pointnet_pp_8dir.py + soft-label cross entropy (train_8dir_KL.py), N=2048, 32 clouds per GPU "
                "(BASELINE: global batch 256 over 8 GPUs), fwd+loss+bwd+Adam, random-init weights (seed 42), device-side centre sampling")
        metric = "clouds/sec fwd+bwd, pointnet_pp_8dir N=2048"
        flops_per_cloud = bench.FLOPS_PER_CLOUD     # the grouped sizes (npoint, nsample) do not depend on N
        bytes_per_cloud = bench.BYTES_PER_CLOUD + 12.0 * (N - 1024)
    opt = optim.FlatAdam(model.parameters(), lr=1e-3)
    graphed = GraphedStep(opt, loss_fn, inputs, adopt_inputs=True, zero_grad_in_graph=False)

    def step():
        loss = graphed(*inputs)
        if clip is not None:
            opt.clip_grad_norm_(clip, return_norm=False)
        opt.step(zero_grad=True)
        return loss

    def eager_step():
        opt.zero_grad()
        loss = loss_fn(*inputs)
        if loss.requires_grad:
            loss.backward()
        if clip is not None:
            opt.clip_grad_norm_(clip, return_norm=False)
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    roof, table, kernel_ms = bench.roofline_leg(eager_step)
    per_gpu = B * args.steps / el
    print(json.dumps({
        "metric": metric, "value": per_gpu, "unit": "clouds/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "config": {"workload": what, "per_gpu_batch": B, "points": N,
                                        "launch": "hipGraph(fwd+loss+bwd) + eager " + ("sumsq + clipped Adam" if clip else "Adam")},
        "final_loss": float(loss.detach()),
        "mfma_fraction": per_gpu * flops_per_cloud / (bench.MFMA_F32_PEAK_TFLOPS * 1e12),
        "hbm_fraction": per_gpu * bytes_per_cloud / (bench.HBM_PEAK_GBS * 1e9),
        "kernel_ms_per_step": kernel_ms, "roofline": roof, "top_kernels": table}))


if __name__ == "__main__":
    main()
