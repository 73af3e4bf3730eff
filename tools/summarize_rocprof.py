#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into the small files that are committed under profiles/.

  tools/summarize_rocprof.py stats <dir> <out.csv>            # --kernel-trace --stats run: per-kernel summary
  tools/summarize_rocprof.py pmc <fetch_dir> <write_dir> <out.json> [<out.csv>]
        # two --pmc passes (FETCH_SIZE, WRITE_SIZE) -> HBM bytes per launch, keyed like bench.py's roofline tags
  tools/summarize_rocprof.py pmc-all <fetch_dir> <write_dir> <out.json> [<out.csv>]
        # the same for EVERY kernel of the run (index kernels, pooling, reductions ...), keyed '<kernel> grid=<wgs>x<threads>'
  Both stamp the output with the git revision and the hash of csrc/ the counters were taken with.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950 FETCH_SIZE
reports half of the bytes of a wide coalesced read stream (MI355X_MICROARCH.md, "HBM"; cdna_hip_programming.md
section 7) -- the x2 is that guide's correction, WRITE_SIZE needs none for 16-byte streaming stores.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def find(d, pat):
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not hits:
        raise SystemExit(f"no {pat} under {d}")
    return hits[0]


def tag_of(name, gx, gy, wg):
    """bench.py-style key "<kernel and template arguments> grid=<workgroups>"; None for kernels without
    a cost model.  bench.py builds the same key from its launch tags (kernel part + grid part)."""
    g = f"grid={(gx // wg) * gy}"   # total workgroups (the counter CSV only has the flattened grid size)
    m = re.search(r"gemm_ws_kernel<(\d+), (\d+), (\d+), \d+, \d+, (\d+), (\d+), (true|false)>", name)
    if m:
        dw = ",dW" if m.group(6) == "true" else ""
        return f"gemm_ws_kernel<{m.group(1)},{m.group(2)},{m.group(3)},A{m.group(4)},E{m.group(5)}{dw}> {g}"
    # the wave-private / folded kernels of round 3 (bench.py tags: gemm_wsp_kernel<K,A5>, gemm_wsq_kernel<K,A5>, gemm_wsx_kernel<K,wpc>,
    # gemm_wsf_kernel<K,NT,Aa,Ee>, gemm_wsf0_kernel<Ee>)
    m = re.search(r"gemm_wsp_kernel<(\d+), (\d+)>", name)
    if m:
        return f"gemm_wsp_kernel<{m.group(1)},A{m.group(2)}> {g}"
    m = re.search(r"gemm_wsq_kernel<(\d+), \d+>", name)
    if m:
        return f"gemm_wsq_kernel<{m.group(1)},A5> {g}"
    m = re.search(r"gemm_wsx_kernel<(\d+), (\d+)(?:, (true|false))?(?:, (true|false))?>", name)   # <K, workgroups per CU, split dA, split dW>
    if m:
        return f"gemm_wsx_kernel<{m.group(1)},{m.group(2)}{',S3' if m.group(3) == 'true' else ''}{',D3' if m.group(4) == 'true' else ''}> {g}"
    m = re.search(r"gemm_wsf_kernel<(\d+), (\d+), (\d+), (\d+)>", name)
    if m:
        return f"gemm_wsf_kernel<{m.group(1)},{m.group(2)},A{m.group(3)},E{m.group(4)}> {g}"
    # round 4: the split-product kernels (bench.py tags: gemm_wsf3_kernel<K,Aa,Ee>, gemm_wsd3_kernel<K,BN,Aa>)
    m = re.search(r"gemm_wsf3_kernel<(\d+), \d+, \d+, (\d+), (\d+)>", name)   # <K, waves, column tiles per wave, A mode, E mode>
    if m:
        return f"gemm_wsf3_kernel<{m.group(1)},A{m.group(2)},E{m.group(3)}> {g}"
    m = re.search(r"gemm_wsd3_kernel<(\d+), (\d+), (\d+)>", name)   # <K, BN, A mode>
    if m:
        return f"gemm_wsd3_kernel<{m.group(1)},{m.group(2)},A{m.group(3)}> {g}"
    m = re.search(r"gemm_wsf03_kernel<(\d+)>", name)
    if m:
        return f"gemm_wsf03_kernel<E{m.group(1)}> {g}"
    m = re.search(r"gemm_wsf0_kernel<(\d+)>", name)
    if m:
        return f"gemm_wsf0_kernel<E{m.group(1)}> {g}"
    m = re.search(r"gemm_smallm_kernel<(\d+), (\d+), (true|false)>", name)
    if m:
        return f"gemm_smallm_kernel<A{m.group(1)},E{m.group(2)},T{1 if m.group(3) == 'true' else 0}> {g}"
    m = re.search(r"dw_kernel<(\d+), (\d+), \d+, \d+>", name)
    if m:
        return f"dw_kernel<A{m.group(1)},A{m.group(2)}> {g}"
    m = re.search(r"gemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", name)
    if m:
        return f"gemm_kernel<{m.group(1)},{m.group(2)},{m.group(3)},{m.group(4)},A{m.group(5)},E{m.group(6)}> {g}"
    return None


def short(name):
    """Kernel name without namespace, return type and parameter list (template arguments kept)."""
    n = re.sub(r"\(.*", "", name)
    n = re.sub(r"^void ", "", n).replace("pnpp::", "")
    return n[:96]


def any_tag(name, gx, gy, wg):
    """Every kernel gets a key: the roofline-style tag where one exists, else '<kernel> grid=<workgroups>x<threads>'."""
    return tag_of(name, gx, gy, wg) or f"{short(name)} grid={(gx // max(wg, 1)) * gy}x{wg}"


def stats(d, out):
    rows = list(csv.DictReader(open(find(d, "*kernel_stats.csv"))))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for r in rows:
            name = re.sub(r"\(.*", "", r["Name"])[:110]
            w.writerow([name, r["Calls"], f"{float(r['TotalDurationNs']) / 1e3:.1f}", f"{float(r['AverageNs']) / 1e3:.2f}",
                        r["Percentage"]])
    # per (kernel, grid) averages from the trace: the same instantiation runs with several shapes
    trace = list(csv.DictReader(open(find(d, "*kernel_trace.csv"))))
    agg = defaultdict(lambda: [0, 0.0])
    for r in trace:
        wg = int(r.get("Workgroup_Size_X", 256) or 256)
        t = tag_of(r["Kernel_Name"], int(r["Grid_Size_X"]), int(r.get("Grid_Size_Y", 1) or 1), wg)
        if t:
            a = agg[t]
            a[0] += 1
            a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    with open(out.replace(".csv", "_by_shape.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel_and_grid", "calls", "avg_us"])
        for t, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([t, n, f"{us / n:.2f}"])
    # every kernel of the run (not only the ones with a cost model), per (kernel, grid): the same instantiation is
    # launched with several shapes per step, and the launch-floor tail only shows when they are told apart
    allk = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for r in trace:
        wg = int(r.get("Workgroup_Size_X", 256) or 256)
        t = any_tag(r["Kernel_Name"], int(r["Grid_Size_X"]), int(r.get("Grid_Size_Y", 1) or 1), wg)
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = allk[t]
        a[0] += 1
        a[1] += us
        a[2] = min(a[2], us)
        a[3] = max(a[3], us)
    total = sum(a[1] for a in allk.values())
    with open(out.replace(".csv", "_all_by_shape.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel_and_grid", "calls", "avg_us", "min_us", "max_us", "percent_of_kernel_time"])
        for t, (n, us, lo, hi) in sorted(allk.items(), key=lambda kv: -kv[1][1]):
            w.writerow([t, n, f"{us / n:.2f}", f"{lo:.2f}", f"{hi:.2f}", f"{100 * us / total:.2f}"])


def pmc(fetch_dir, write_dir, out_json, out_csv=None, every_kernel=False, meta=None):
    tagger = any_tag if every_kernel else tag_of

    def load(d, counter):
        res = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(find(d, "*counter_collection.csv"))):
            if r["Counter_Name"] != counter:
                continue
            wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)) or 256)
            gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
            gy = int(r.get("Grid_Size_Y", 1) or 1)
            t = tagger(r["Kernel_Name"], gx, gy, wg)
            if t:
                res[t][0] += 1
                res[t][1] += float(r["Counter_Value"])
        return {t: v[1] / v[0] for t, v in res.items()}
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    out = {}
    for t in sorted(set(fetch) | set(write)):
        f_kib, w_kib = fetch.get(t, 0.0), write.get(t, 0.0)
        out[t] = {"hbm_bytes_per_launch": (2.0 * f_kib + w_kib) * 1024.0, "FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib}
    if meta:
        out["_measured_at"] = meta
    json.dump(out, open(out_json, "w"), indent=1, sort_keys=True)
    out.pop("_measured_at", None)
    if out_csv:
        with open(out_csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel_and_grid", "FETCH_SIZE_KiB_raw", "WRITE_SIZE_KiB", "hbm_MB_per_launch_corrected"])
            for t, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
                w.writerow([t, f"{v['FETCH_SIZE_KiB']:.0f}", f"{v['WRITE_SIZE_KiB']:.0f}", f"{v['hbm_bytes_per_launch'] / 1e6:.2f}"])


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] in ("pmc", "pmc-all"):
        # stamp what the counters belong to: git revision and the hash of the kernel sources (bench.py voids stale counters)
        import hashlib
        import subprocess
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        h = hashlib.sha256()
        cs = os.path.join(root, "3d-pointcloud-orientation-estimation_amd", "csrc")
        for fn in sorted(os.listdir(cs)):
            if fn.endswith((".hip", ".h")):
                h.update(open(os.path.join(cs, fn), "rb").read())
        try:
            rev = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        except Exception:
            rev = ""
        rev = os.environ.get("PNPP_GIT_REV", rev) or "unknown"
        pmc(*sys.argv[2:6], every_kernel=sys.argv[1] == "pmc-all", meta={"git": rev, "csrc_sha256": h.hexdigest()[:16]})
