#!/usr/bin/env python3
"""Per-GPU batch sweep of bench.py's step (32 -> 1024 clouds per GPU, N=1024): one child process per batch
size (fresh allocator, fresh graph), prints one JSON document with clouds/s, ms per step and the fractions of both roofs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rows = []
    for B in (32, 64, 128, 256, 512, 1024):
        steps = max(20, 6400 // B)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", str(steps), "--warmup", "10",
                            "--no-cpu-baseline", "--no-roofline", "--no-mfma-variant"], capture_output=True, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            rows.append({"per_gpu_batch": B, "error": (r.stderr or r.stdout)[-400:]})
            continue
        d = json.loads(line[0])
        rows.append({"per_gpu_batch": B, "clouds_per_s": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                     "mfma_fraction": d["mfma_fraction"], "hbm_fraction": d["hbm_fraction"], "launch": d["config"]["launch"]})
        print(f"[sweep] B={B}: {d['value']:.0f} clouds/s, {d['ms_per_step']:.3f} ms/step", file=sys.stderr)
    print(json.dumps({"workload": "bench.py step, pointnet_pp_vonMises N=1024, one MI355X, float32", "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
