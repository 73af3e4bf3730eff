#!/usr/bin/env python3
"""Gaps between consecutive kernels of the replayed step, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace -d <dir> -o t --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-bf16-variant
    python tools/step_gaps.py <dir>
(last 100 steps; a step ends with the Adam launch)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
# take the last 100 steps' worth: find adam kernels as step delimiters
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
lo, hi = idx[-101], idx[-1]
seg = rows[lo:hi + 1]
gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
busy = sum(e - s for s, e, _ in seg[1:])
steps = 100
print("per step: kernels %.1f us, gaps %.1f us (%d launches), wall %.1f us" % (busy / steps / 1e3, sum(gaps) / steps / 1e3, len(seg) // steps, (seg[-1][1] - seg[0][1]) / steps / 1e3))
by = collections.defaultdict(list)
for i, g in enumerate(gaps): by[(seg[i][2][:28], seg[i + 1][2][:28])].append(g)
top = sorted(by.items(), key=lambda kv: -sum(kv[1]) / steps)[:12]
for (a, b), v in top: print("%6.2f us/step  n=%4d  %s -> %s" % (sum(v) / steps / 1e3, len(v), a, b))
