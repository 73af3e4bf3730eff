#!/usr/bin/env python3
"""Script-level throughput of the drop-in trainer (what a user of train_single_peak_vonMises_KL.py gets), eager vs the
hipGraph path, next to bench.py's number:  python tools/script_throughput.py [--clouds 8192] [--epochs 4]

Runs the script's own main() on generated clouds (N=1024, batch 32, device-side centre sampler) and reports training-phase
clouds/s of the last epoch (per-sample losses are read back once per phase, which ends the timed span)."""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3d-pointcloud-orientation-estimation_amd")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=8192)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--script", default="train_single_peak_vonMises_KL")
    args = ap.parse_args()
    os.environ.update(PNPP_NUM_POINTS="1024", PNPP_BATCH="32", PNPP_EPOCHS=str(args.epochs),
                      PNPP_RES=tempfile.mkdtemp(prefix="pnpp_res_"))
    mod = __import__(args.script)
    out = {}
    for mode, env in (("graph", "0"), ("eager", "1")):
        os.environ["PNPP_NO_GRAPH"] = env
        hist, _ = mod.main(["--synthetic", str(args.clouds), "--sampler", "device"])
        h = hist if "seconds" in hist else hist["_trainer"]   # the multi-peak script returns per-category curves + "_trainer"
        sec, n = h["seconds"]["train"][-1], h["samples"]["train"][-1]
        out[mode] = {"train_clouds_per_s": n / sec, "train_seconds_last_epoch": sec, "clouds": n, "steps": h["steps"]}
    print(json.dumps({"script": args.script, "points": 1024, "batch": 32, **out}))


if __name__ == "__main__":
    main()
