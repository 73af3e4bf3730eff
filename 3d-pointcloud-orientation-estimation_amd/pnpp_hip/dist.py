"""One process per GPU, batch (cloud) sharding, ONE gradient exchange per step.

The reference is single-process; this is the MI355X-native scale-out: every rank holds a
replica, takes a contiguous shard of the global batch, and after backward the ranks sum one flat float32
gradient buffer (~5.9 MB) with a single all-reduce -- RCCL over xGMI on GPUs (`backend="nccl"` is RCCL on
ROCm), gloo in the CPU tests.  The 1/world factor is folded into the optimiser's grad_scale, BatchNorm
statistics stay per rank (what DDP does to the reference's modules), and centre sampling is decorrelated
across ranks through pnpp_hip.sampling.set_rank.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialises torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* when WORLD_SIZE > 1."""
    rank, local_rank, world = env_world()
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # PNPP_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsals on a single-GPU box; RCCL refuses that)
        backend = backend or os.environ.get("PNPP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    try:
        from . import sampling
        sampling.set_rank(rank)
    except Exception:  # pragma: no cover
        pass
    return rank, local_rank, world


def shard_bounds(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of the global batch for `rank` (first `global_batch % world` ranks get one more)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_flat(flat: torch.Tensor, src: int = 0) -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src)


def all_reduce_flat_grad(flat_g: torch.Tensor, async_op: bool = False):
    """Sum the flat gradient buffer over ranks (the caller folds 1/world into the optimiser step)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        return dist.all_reduce(flat_g, op=dist.ReduceOp.SUM, async_op=async_op)
    return None


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


# ------------------------------------------------------------------------------------------------
# SyncBN (off by default): train-mode BatchNorm statistics summed over the ranks
# ------------------------------------------------------------------------------------------------
_sync_state = {"cb": None, "buf": None, "calls": 0}


def sync_batchnorm_enabled() -> bool:
    return _sync_state["cb"] is not None


def sum_over_ranks(t: torch.Tensor) -> torch.Tensor:
    """In-place sum of `t` over the ranks (identity in a single process): what a BatchNorm layer does with its
    (sum, sum of squares, row count) forward and (sum dy, sum dy*xhat, row count) backward under SyncBN.  The HIP library's
    exchange callback goes through here (and so do the tests' CPU checks of the same exchange)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def enable_sync_batchnorm(max_channels: int = 4096) -> None:
    """torch.nn.SyncBatchNorm.convert_sync_batchnorm for this path: from now on every training-mode BatchNorm of the HIP
    library exchanges its per-channel sums over the ranks (pnpp_set_stats_exchange), so a global batch split over the ranks
    normalises like the single process on the concatenated batch (the reference's nn.BatchNorm2d / BatchNorm1d,
    models/pointnet_pp_8dir.py:40-41).  22 small all-reduces per step for PointNetPPVonMises; the BatchNorm head takes its
    unfused route and dropout masks are drawn by torch (ops.fc_block).  Gradients of the BatchNorm parameters stay rank-local
    sums: average them with the usual flat-gradient all-reduce."""
    from . import _lib as L
    if _sync_state["cb"] is not None:
        return
    dev = torch.device("cuda", torch.cuda.current_device())
    buf = torch.zeros(2 * (2 * int(max_channels) + 1), device=dev, dtype=torch.float64)

    def _exchange(ptr, n, stream, user):
        try:
            if ptr != buf.data_ptr() or n > buf.numel() // 2:
                return -2
            if stream is not None and int(stream or 0) != int(torch.cuda.current_stream().cuda_stream or 0):
                return -3   # the library enqueues on the stream it was given: it must be torch's current one
            sum_over_ranks(buf[:n])
            _sync_state["calls"] += 1
            return 0
        except Exception as e:  # an exception must not cross the C ABI
            print(f"[pnpp_hip.dist] statistics exchange failed: {type(e).__name__}: {e}", flush=True)
            return -1

    cb = L.STATS_EXCHANGE_FN(_exchange)
    L.check(L.lib().pnpp_set_stats_exchange(cb, None, buf.data_ptr(), buf.numel()))
    _sync_state["cb"], _sync_state["buf"] = cb, buf          # keep both alive: the library holds raw pointers


def disable_sync_batchnorm() -> None:
    from . import _lib as L
    if _sync_state["cb"] is None:
        return
    L.check(L.lib().pnpp_set_stats_exchange(L.STATS_EXCHANGE_FN(0), None, None, 0))
    _sync_state["cb"], _sync_state["buf"] = None, None
