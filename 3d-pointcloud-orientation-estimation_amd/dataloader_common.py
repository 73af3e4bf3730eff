"""Shared PLY / sampling helpers of the three drop-in dataloader modules.

Reference: dataloader_single_peak_vonMises.py:5-14, dataloader_multi_peak_vonMises.py:6-26,
dataloader_8dir_sampled.py:6-16 (three copies of the same two functions).

The reference parses every 10k-line ASCII PLY with np.loadtxt on every __getitem__; at the rates the GPU
path sustains that is three to four orders of magnitude too slow.  read_ply therefore keeps an
optional binary side cache: set PNPP_PLY_CACHE=1 to write `<file>.ply.npy` next to the source (or
PNPP_PLY_CACHE_DIR=<dir> for a separate directory) the first time a file is parsed and memory-map it afterwards.
The parsed values are identical (float32 of the first three columns).
"""
import hashlib
import os

import numpy as np


def _cache_path(p):
    d = os.environ.get("PNPP_PLY_CACHE_DIR")
    if d:
        os.makedirs(d, exist_ok=True)
        return os.path.join(d, hashlib.sha1(os.path.abspath(str(p)).encode()).hexdigest() + ".npy")
    if os.environ.get("PNPP_PLY_CACHE") == "1":
        return str(p) + ".npy"
    return None


def _parse_ascii_ply(p):
    with open(p, "r") as f:
        while True:
            line = f.readline()
            if not line or line.strip() == "end_header":
                break
        body = f.read()
    rows = [ln.split() for ln in body.splitlines() if ln.strip()]
    if not rows:
        return np.zeros((0, 3), np.float32)
    width = min(len(r) for r in rows)
    if width < 3:
        raise ValueError(f"PLY vertex rows of {p} have fewer than three columns")
    flat = np.array([r[:3] for r in rows], dtype=np.float32)
    return flat


def read_ply(p):
    """ASCII PLY -> (n,3) float32 array of the first three vertex properties (x y z)."""
    c = _cache_path(p)
    if c and os.path.exists(c) and os.path.getmtime(c) >= os.path.getmtime(p):
        return np.load(c, mmap_mode="r")
    pts = _parse_ascii_ply(p)
    if c:
        np.save(c, pts)
    return pts


def sample_pts(arr, num=10_000):
    """`num` points of `arr`: without replacement when enough points exist, with replacement otherwise
    (np.random global generator, as in the reference)."""
    n = len(arr)
    if n == 0:
        return np.asarray(arr)
    idx = np.random.choice(n, num, replace=n < num)
    return np.asarray(arr)[idx]
