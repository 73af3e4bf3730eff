#!/usr/bin/env python3
"""One line per kernel matching a filter from a kernel_stats_all_by_shape.csv: tools/ab_show.py <csv> [filter[,filter..]]"""
import csv
import sys
flt = sys.argv[2].split(",") if len(sys.argv) > 2 else [""]
tot = 0.0
out = []
for row in csv.DictReader(open(sys.argv[1])):
    n = int(row["calls"]) / 223.0
    if n < 0.9:
        continue
    tot += float(row["avg_us"]) * n
    if any(f in row["kernel_and_grid"] for f in flt):
        out.append(f'{float(row["avg_us"]):6.1f} {row["kernel_and_grid"][:46]}')
print("   sum_us %.1f | " % tot + " | ".join(out))
