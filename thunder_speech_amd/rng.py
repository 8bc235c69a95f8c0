"""Seeds for the counter-based (Philox) device kernels: dither, dropout, device-side SpecAugment geometry.

A kernel's random values are a pure function of (seed, element index), so all the host has to provide per call is one
63-bit seed.  It is drawn from torch's CPU generator: `torch.manual_seed(...)` therefore makes a training run
reproducible, and no device synchronisation is involved."""
from __future__ import annotations

import torch


def next_seed() -> int:
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
