"""TEST INFRASTRUCTURE (CPU oracle) -- SpecAugment / SpecCutout (reference quartznet/spec_augment.py:23-102).

The mask arithmetic is torchaudio.functional.mask_along_axis (torchaudio 0.12.0, absent from this image; the reference's own
`_create_mask`, spec_augment.py:60-75, is its published copy): value = rand(1) * mask_param; min_value = rand(1) * (size - value);
start = long(min_value); end = long(min_value) + long(value); the SAME mask for every clip of the batch.  `draw_*` take the
uniform draws as arguments so that the same function serves the torch-generator path (reference-exact) and the Philox path."""
from __future__ import annotations

from typing import Callable, List, Tuple

import numpy as np
import torch

from . import philox as ph


def span(u_value: float, u_min: float, mask_param: int, size: int) -> Tuple[int, int]:
    value = np.float32(u_value) * np.float32(mask_param)
    min_value = np.float32(u_min) * (np.float32(size) - value)
    return int(min_value), int(min_value) + int(value)


def draw_table(rand2: Callable[[], Tuple[float, float]], n_mels: int, n_frames: int, n_time=0, time_width=0, n_freq=0, freq_width=0,
               n_cutout=0, cut_time_width=0, cut_freq_width=0) -> np.ndarray:
    """Rows (f0, f1, t0, t1) in the order the reference applies them: cutout rectangles (frequency span, then a time span drawn
    with cut_FREQ_width -- spec_augment.py:99-100), time masks, frequency masks (spec_augment.py:51-56)."""
    rows: List[Tuple[int, int, int, int]] = []
    for _ in range(n_cutout):
        f0, f1 = span(*rand2(), cut_freq_width, n_mels)
        t0, t1 = span(*rand2(), cut_freq_width, n_frames)
        rows.append((f0, f1, t0, t1))
    for _ in range(n_time):
        t0, t1 = span(*rand2(), time_width, n_frames)
        rows.append((0, n_mels, t0, t1))
    for _ in range(n_freq):
        f0, f1 = span(*rand2(), freq_width, n_mels)
        rows.append((f0, f1, 0, n_frames))
    return np.asarray(rows, dtype=np.int32).reshape(-1, 4)


def torch_rand2():
    """The reference's host draws: torch.rand(1) for `value`, then torch.rand(1) for `min_value` (CPU default generator)."""
    a = float(torch.rand(1))
    b = float(torch.rand(1))
    return a, b


def philox_rand2(seed: int):
    state = {"ctr": 0}

    def draw():
        r = ph.philox(seed, ph.SPEC, np.asarray([state["ctr"]], dtype=np.uint64))
        state["ctr"] += 1
        return float(ph.u01(r[0])[0]), float(ph.u01(r[1])[0])
    return draw


def apply_table(x: torch.Tensor, table: np.ndarray) -> torch.Tensor:
    """x [B, F, T] -> masked copy (masked_fill(mask, 0.0) per row)."""
    y = x.clone()
    for f0, f1, t0, t1 in table.tolist():
        y[:, max(f0, 0):f1, max(t0, 0):t1] = 0.0
    return y
