"""Builds libpnpp_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.

The library is built IN-TREE (next to this file) so that it travels with a source snapshot;
nothing is installed into site-packages.  hipcc cross-compiles without a GPU present.

    python -m pnpp_hip.build          # from 3d-pointcloud-orientation-estimation_amd/
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libpnpp_hip.so")

ARCH = "gfx950"
COMMON = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-Wno-unused-value", "-Wno-pass-failed"]
# per-file extra flags: the index kernels pin the reference's float32 rounding sequence, so the
# compiler must not introduce fused multiply-adds of its own there
# the split-product kernels: packed float32 vector instructions (what the SLP vectoriser makes of the operand splits) issue slower beside
# MFMAs than the scalar forms (measured: 47.9 -> 42.1 us on the sa1 launch)
_NOSLP = [] if os.environ.get("PNPP_SLP") else ["-fno-slp-vectorize"]
SOURCES = {
    "index_kernels.hip": ["-ffp-contract=off"],
    "gemm_kernels.hip": (["-DPNPP_STAMPS"] if os.environ.get("PNPP_STAMPS") else []),
    "gemm_bf16_kernels.hip": [],
    "gemm_mid_kernels.hip": (["-DMID3_STAMPS"] if os.environ.get("PNPP_STAMPS") else []),
    "gemm_wsf_kernels.hip": (["-DPNPP_STAMPS"] if os.environ.get("PNPP_STAMPS") else []) + ([f"-DWSF_EXP={os.environ['PNPP_WSF_EXP']}"] if os.environ.get("PNPP_WSF_EXP") else []),
    "gemm_wsf3_kernels.hip": _NOSLP,
    "gemm_wsd3_kernels.hip": _NOSLP + (["-DPNPP_STAMPS"] if os.environ.get("PNPP_STAMPS") else []),
    "gemm_wsp_kernels.hip": (["-DPNPP_STAMPS"] if os.environ.get("PNPP_STAMPS") else []),
    "gemm_wsx_kernels.hip": [],
    "gemm_wsf03_kernels.hip": _NOSLP,
    "gemm_wsq_kernels.hip": [],
    "loss_kernels.hip": (["-DPNPP_STAMPS"] if os.environ.get("PNPP_STAMPS") else []),
    "sa_api.hip": [],
    "fc_api.hip": [],
    "transformer_kernels.hip": [],
}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libpnpp_hip.so cannot be built")
    return exe


def _deps_mtime() -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(INCLUDE, "pnpp_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile what is out of date and link.  Returns the path of the shared library."""
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdr_t = _deps_mtime()
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc, *COMMON, *extra, "-c", s, "-o", o]
        stale_flags = not os.path.exists(o + ".cmd") or open(o + ".cmd").read() != " ".join(cmd)   # an object belongs to its command line
        if force or stale_flags or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t):
            jobs.append(cmd)

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stderr[-4000:])
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)
        if "-c" in cmd:
            with open(cmd[-1] + ".cmd", "w") as f:
                f.write(" ".join(cmd))

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
