"""Device-side centre sampling (throughput mode of models/pointnet_pp_8dir.py:28).

`torch.manual_seed(s)` still controls the draw: the kernel's Philox key is the CPU generator's
initial seed, its counter carries (rank, call number), so every call -- and every rank under data
parallelism -- gets an independent, reproducible stream without touching the host generator."""
import torch

from . import ops

_state = {"calls": 0, "rank": 0}


def set_rank(rank: int) -> None:
    _state["rank"] = int(rank)


def reset(calls: int = 0) -> None:
    _state["calls"] = int(calls)


def device_random_centres(B: int, N: int, npoint: int, device) -> torch.Tensor:
    _state["calls"] += 1
    stream_id = (_state["rank"] << 40) | _state["calls"]
    return ops.sample_random(torch.initial_seed(), stream_id, B, N, npoint, device)
