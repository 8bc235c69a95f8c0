"""Multi-GPU helpers.  Inference shards by clip: every clip is an independent unit (eval-mode BatchNorm uses
running statistics; the reference tests batch independence in tests/utils.py:70-97), so each rank owns a
contiguous slice of the clips and there is NO collective on the data path.  The only collectives are the
rendezvous barrier and the max-over-ranks reduction of the measured time (bench.py)."""
from __future__ import annotations

from typing import Tuple

import torch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) slice of `n_items` for `rank` (first n % world ranks get one extra)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def max_over_ranks(seconds: float, device=None) -> float:
    """MAX all-reduce of a host-side duration (RCCL when `device` is a GPU, gloo on CPU)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
