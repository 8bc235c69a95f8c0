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
