"""Seeds for the counter-based (Philox) device kernels: dither, dropout, device-side SpecAugment geometry.

A kernel's random values are a pure function of (seed, element index), so all the host has to provide per call is one
63-bit seed.  It is drawn from torch's CPU generator: `torch.manual_seed(...)` therefore makes a training run
reproducible, and no device synchronisation is involved."""
from __future__ import annotations

import torch


def next_seed() -> int:
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


_NONCE = {}


def replay_nonce(device) -> torch.Tensor:
    """Device-side uint64 (stored as int64 [1]) every dropout kernel adds to its by-value seed.  Zero outside hipGraphs; a graphed
    training step (train_graph.GraphedTrainStep) captures `bump_replay_nonce` at its head, so each replay of the frozen launch
    arguments still draws fresh masks."""
    key = str(torch.device(device))
    if key not in _NONCE:
        _NONCE[key] = torch.zeros(1, dtype=torch.int64, device=device)
    return _NONCE[key]


def bump_replay_nonce(device) -> None:
    from . import _lib
    n = replay_nonce(device)
    _lib.check(_lib.lib().ts_counter_add(n.data_ptr(), 0x9E3779B97F4A7C15, torch.cuda.current_stream(n.device).cuda_stream), "ts_counter_add")
