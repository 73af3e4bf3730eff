#!/usr/bin/env python3
"""Per-kernel A/B table from tools/ab_trace.sh runs: tools/ab_diff.py <tag> <rounds>  (averages kernel_stats_all_by_shape.csv)."""
import csv
import sys

tag, rounds = sys.argv[1], int(sys.argv[2])


def load(v):
    acc = {}
    for r in range(1, rounds + 1):
        with open(f"gpurun_out/{tag}_{v}{r}/kernel_stats_all_by_shape.csv") as f:
            for row in csv.DictReader(f):
                if int(row["calls"]) < 200:
                    continue
                k = row["kernel_and_grid"]
                a = acc.setdefault(k, [0.0, 0, 0.0])
                a[0] += float(row["avg_us"])
                a[1] += 1
                a[2] = int(row["calls"]) / 223.0
    return {k: (v[0] / v[1], v[2]) for k, v in acc.items()}


A, B = load("A"), load("B")
tot = 0.0
rows = []
for k in sorted(set(A) | set(B)):
    a, b = A.get(k, (0, 0)), B.get(k, (0, 0))
    d = (b[0] - a[0]) * max(a[1], b[1])
    tot += d
    rows.append((d, k, a[0], b[0], max(a[1], b[1])))
for d, k, a, b, n in sorted(rows):
    if abs(d) >= 0.15:
        print(f"{d:+7.2f} us/step  {a:7.2f} -> {b:7.2f} x{n:.0f}  {k}")
print(f"total {tot:+.2f} us per step (B - A); A sum {sum(v[0] * v[1] for v in A.values()):.1f}  B sum {sum(v[0] * v[1] for v in B.values()):.1f}")
