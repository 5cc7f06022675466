"""Multi-GPU layout of the env batch: independent contiguous shards, one process per GPU, no collective
on the data path (each env is its own world in the reference: one Bullet client per env, env_base.py:55).
torch.distributed is used only for the start/stop barrier and the MAX of the per-rank times.
"""
from __future__ import annotations

from typing import Tuple


def env_range(rank: int, world: int, envs_per_rank: int) -> Tuple[int, int]:
    """Global env ids [lo, hi) owned by `rank` under weak scaling (fixed envs per GPU)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return rank * envs_per_rank, (rank + 1) * envs_per_rank


def max_over_ranks(seconds: float, dist=None, device=None) -> float:
    """Slowest rank's time; with dist=None (single process) returns the input."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(values, dist=None):
    """[world][len(values)] floats, rank order: every rank's own numbers (per-rank step time, kernel time) for the report."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [[float(v) for v in values]]
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o] for o in out]


def gather_objects(obj, dist=None):
    """[world] picklable objects, rank order (each rank's device identity for the report)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def aggregate_throughput(envs_per_rank: int, world: int, steps: int, seconds: float) -> float:
    """Whole-job env-steps/s: all ranks' envs x steps over the slowest rank's time."""
    return envs_per_rank * world * steps / seconds
