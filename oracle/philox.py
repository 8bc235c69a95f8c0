"""TEST INFRASTRUCTURE (CPU oracle) -- Philox4x32-10 and the uniform / normal transforms of csrc/ts_philox.hpp, restated in
numpy integer arithmetic (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the reference uses torch's
generators, which are not reproducible across devices, so the device stream is pinned against THIS restatement and the
reference's semantics are checked on mask statistics and explicit-mask application instead).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
from __future__ import annotations

import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
DITHER, SPEC, DROPOUT = 1, 2, 3        # stream ids (csrc/ts_philox.hpp)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """All arguments uint32 numpy arrays (broadcastable); returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) & 0xFFFFFFFF for v in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c0, np.uint64(M1) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & np.uint64(0xFFFFFFFF)
        n1 = p1 & np.uint64(0xFFFFFFFF)
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & np.uint64(0xFFFFFFFF)
        n3 = p0 & np.uint64(0xFFFFFFFF)
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(W0)) & np.uint64(0xFFFFFFFF)
        k1 = (k1 + np.uint64(W1)) & np.uint64(0xFFFFFFFF)
    return tuple(v.astype(np.uint32) for v in (c0, c1, c2, c3))


def philox(seed: int, stream: int, counter):
    counter = np.asarray(counter, dtype=np.uint64)
    return philox4x32_10(counter & np.uint64(0xFFFFFFFF), counter >> np.uint64(32), np.uint64(stream), np.uint64(0),
                         seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def u01(x):
    return (np.asarray(x, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def normal2(x0, x1):
    u1 = ((np.asarray(x0, dtype=np.uint32) >> np.uint32(8)).astype(np.float64) + 1.0) * 2.0 ** -24
    r = np.sqrt(-2.0 * np.log(u1))
    th = 2.0 * np.pi * u01(x1).astype(np.float64)
    return (r * np.cos(th)).astype(np.float32), (r * np.sin(th)).astype(np.float32)


def dither_noise(seed: int, b: int, n_samples: int) -> np.ndarray:
    """Noise of samples 0..n_samples-1 of clip b (csrc/frontend.hip dither_noise): counter = b << 32 | k >> 1."""
    k = np.arange(n_samples, dtype=np.uint64)
    ctr = (np.uint64(b) << np.uint64(32)) | (k >> np.uint64(1))
    r = philox(seed, DITHER, ctr)
    n0, n1 = normal2(r[0], r[1])
    return np.where((k & np.uint64(1)) == 1, n1, n0).astype(np.float32)


def dropout_keep(seed: int, n: int, p: float) -> np.ndarray:
    """Keep mask of csrc/augment.hip dropout_kernel: element e uses word e & 3 of counter e >> 2."""
    g = np.arange((n + 3) // 4, dtype=np.uint64)
    r = philox(seed, DROPOUT, g)
    u = np.stack([u01(v) for v in r], axis=1).reshape(-1)[:n]
    return u >= np.float32(p)
