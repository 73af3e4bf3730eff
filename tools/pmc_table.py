#!/usr/bin/env python3
"""Per-kernel averages of arbitrary rocprofv3 --pmc counters: tools/pmc_table.py <dir> [name-filter]"""
import csv, glob, os, re, sys
from collections import defaultdict
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void pnpp::", "")
    if flt and flt not in name:
        continue
    key = f"{name} g={int(r['Grid_Size']) // int(r['Workgroup_Size'])}"
    a = agg[key][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
ctrs = sorted({c for v in agg.values() for c in v})
print("kernel | " + " | ".join(ctrs))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1][ctrs[0]])[1]):
    print(k[:70], "|", " | ".join(f"{v[c][1] / max(v[c][0], 1):.3g}" for c in ctrs))
