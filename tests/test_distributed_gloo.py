"""world_size-2 gloo test of the N>1 path used by bench.py: clip sharding (no data-path collective) and the
max-over-ranks reduction of the measured time."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from thunder_speech_amd.parallel import max_over_ranks, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 65, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, port, n_clips, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b, e = shard_range(n_clips, rank, world)
        owned = torch.zeros(n_clips, dtype=torch.int64)
        owned[b:e] = 1
        dist.all_reduce(owned)                       # test-only collective: every clip owned exactly once
        slow = max_over_ranks(1.0 + rank)            # the slowest rank defines the step time
        dist.barrier()
        if rank == 0:
            out.put((owned.tolist(), slow))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_time_reduction():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 65, out)) for r in range(2)]
    for p in procs:
        p.start()
    owned, slow = out.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert owned == [1] * 65
    assert slow == 2.0


def _grad_worker(rank, world, port, out):
    from thunder_speech_amd.parallel import allreduce_gradients
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)                   # identical "weights" on every rank (replicas)
        params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in ((29, 64, 1), (29,), (5, 7), (1000,))]
        params.append(torch.nn.Parameter(torch.zeros(3)))       # a parameter without a gradient is skipped
        for i, p in enumerate(params[:-1]):
            p.grad = torch.full_like(p, float(rank + 1)) * (i + 1)
        n_small = allreduce_gradients(params, bucket_bytes=4096)     # forces several buckets
        small = [p.grad.clone() for p in params[:-1]]
        for i, p in enumerate(params[:-1]):
            p.grad = torch.full_like(p, float(rank + 1)) * (i + 1)
        n_big = allreduce_gradients(params)                          # one bucket
        dist.barrier()
        if rank == 0:
            out.put((n_small, n_big, [float(s.mean()) for s in small], [float(p.grad.mean()) for p in params[:-1]],
                     params[-1].grad is None))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_allreduce_averages_in_buckets():
    """The one exchange step of data-parallel fine-tuning: mean of the replicas' gradients."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    n_small, n_big, small, big, untouched = out.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert n_big == 1 and n_small > 1 and untouched
    want = [1.5 * (i + 1) for i in range(4)]                   # mean of (rank + 1) * (i + 1) over ranks 0, 1
    assert small == pytest.approx(want) and big == pytest.approx(want)
