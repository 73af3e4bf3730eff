"""A line in bench.py's format (metric / value / roofline) for a workload other than the driver's: the per-kernel event rows of the
library's profiler go through bench.py's own cost models (kernel_cost / dominant_roofline), so the `roofline` object means the same
as on the bench line.  `traffic` is filled only for kernels profiles/pmc_traffic.json holds (the bench step's own)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def profiled_rows(step, nsteps):
    """(tag, launches, total ms) per kernel over `nsteps` instrumented calls of step()."""
    import torch
    from pnpp_hip import _lib
    lib = _lib.lib()
    torch.cuda.synchronize()
    lib.pnpp_profile_enable(1)
    for _ in range(nsteps):
        step()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    rc = lib.pnpp_profile_report(buf, len(buf))
    lib.pnpp_profile_enable(0)
    if rc < 0:
        raise RuntimeError("pnpp_profile_report: " + _lib.last_error())
    rows = []
    for line in buf.value.decode().splitlines():
        tag, cnt, ms = line.split("\t")
        rows.append((tag, int(cnt), float(ms)))
    return sorted(rows, key=lambda r: -r[2])


def line(workload, clouds_per_step, sec_per_step, steps, warmup, rows, nsteps_profiled, **extra):
    import bench
    rec = {"metric": "clouds/sec fwd+bwd", "value": clouds_per_step / sec_per_step, "unit": "clouds/s", "n_gpus": 1, "steps": steps,
           "warmup": warmup, "ms_per_step": 1e3 * sec_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic", "config": {"workload": workload},
           "roofline": bench.dominant_roofline(rows, nsteps_profiled),
           "kernel_ms_per_step": sum(r[2] for r in rows) / nsteps_profiled}
    rec.update(extra)
    return rec
