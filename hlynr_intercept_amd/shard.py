"""Sharding of an environment set over the GPUs of one node.

Environments are independent (no env reads another env's state: rl_system/environment.py holds only
`self.*`), so the step needs NO collective: rank r owns the contiguous slab of global env ids
[offset, offset + count), creates its own `hlx_env` with `env_id_offset = offset`, and because the
counter-based RNG is keyed by (seed, GLOBAL env id, vec-step) the union of the shards is bit-identical
to one big env set.  The only collective traffic is the benchmark's barrier + max(time).
"""
from __future__ import annotations

from typing import Tuple


def shard_range(total_envs: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(offset, count) of rank's slab; slabs are contiguous, ordered by rank, sizes differ by at most 1."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    if total_envs < 0:
        raise ValueError("total_envs must be >= 0")
    base, rem = divmod(total_envs, world_size)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """Slowest rank's time (what bounds whole-job throughput).  `dist` = torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value: float, dist=None, device=None):
    """Every rank's value, in rank order (a SCALE record can then show that N ranks really ran and how far apart they were)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def whole_job_throughput(envs_per_rank: int, steps: int, world_size: int, elapsed_max: float) -> float:
    """env-steps/s of the whole job: all ranks' work over the slowest rank's time."""
    return float(envs_per_rank) * steps * world_size / elapsed_max
