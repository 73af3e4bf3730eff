#!/usr/bin/env python3
"""Compact table of hipcc's -Rpass-analysis=kernel-resource-usage remarks (VGPRs, spills, occupancy, LDS).

usage: hipcc ... -Rpass-analysis=kernel-resource-usage file.hip -o /dev/null 2>&1 | tools/kernel_resources.py [filter]
"""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    cur, rows = None, []
    for line in sys.stdin:
        m = re.search(r"Function Name: (\S+)", line) or re.search(r"remark: .*? Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    for r in rows:
        name = demangle(r["name"])
        name = re.sub(r"\(.*", "", name).replace("void pnpp::", "")
        if flt and flt not in name:
            continue
        name = re.sub(r"_ZN4pnpp\d+", "", name); name = re.sub(r"EEvNS_.*", "", name).replace("ELi", ",").replace("ILi", "<").replace("ELb", ",b"); print(f"{name:60s} vgpr={r.get('vgpr', -1):3d} agpr={r.get('agpr', 0):3d} spill={r.get('spill', 0):3d} "
              f"scratch={r.get('scratch', 0):4d} occ={r.get('occ', -1)} lds={r.get('lds', 0)}")


if __name__ == "__main__":
    main()
