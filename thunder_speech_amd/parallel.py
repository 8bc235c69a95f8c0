"""Multi-GPU helpers.

Fine-tuning: data parallel, one process per GPU, replicas hold the full weights; the ONE exchange step per
training step is the gradient average -- `allreduce_gradients` flattens the gradients into a few large buckets (xGMI
is point-to-point: few big collectives beat many small ones) and calls `torch.distributed.all_reduce` (RCCL on the
GPUs, gloo in the CPU tests).  BatchNorm statistics stay per rank, as under the reference's Lightning DDP (quirk A4).

Inference shards by clip: every clip is an independent unit (eval-mode BatchNorm uses
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


def allreduce_gradients(params, bucket_bytes: int = 64 << 20) -> int:
    """Average `p.grad` of every parameter over the ranks, in place; returns the number of collectives issued.
    Buckets are filled in parameter order up to `bucket_bytes` (default 64 MiB: the whole QuartzNet15x5 gradient, 75.7 MB
    fp32, travels in two all-reduces)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    grads = [p.grad for p in params if p.grad is not None]
    n_coll, i = 0, 0
    while i < len(grads):
        j, size = i, 0
        while j < len(grads) and (j == i or size + grads[j].numel() * grads[j].element_size() <= bucket_bytes) \
                and grads[j].dtype == grads[i].dtype and grads[j].device == grads[i].device:
            size += grads[j].numel() * grads[j].element_size()
            j += 1
        flat = torch.cat([g.reshape(-1) for g in grads[i:j]])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        off = 0
        for g in grads[i:j]:
            g.copy_(flat[off: off + g.numel()].view_as(g))
            off += g.numel()
        n_coll += 1
        i = j
    return n_coll
