"""Validation metrics on the device -- what the reference's validation_step (module.py:153-163) gets from torchmetrics'
CharErrorRate / WordErrorRate: sum of Levenshtein distances / sum of reference lengths, over characters / words.

The strings live on the host (they come out of decode_prediction); they are turned into int32 symbol sequences (code
points; word ids from a per-call dictionary), uploaded in one buffer and ALL pairs are scored by one launch of
ts_edit_distance (anti-diagonal dynamic programme, one workgroup per pair).  The running sums stay on the device; nothing
synchronises until `.compute()` is read."""
from __future__ import annotations

from typing import List, Sequence, Union

import numpy as np
import torch

from . import _lib


def _flatten(seqs: Sequence[Sequence[int]]):
    off = np.zeros(len(seqs) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(s) for s in seqs])
    flat = np.fromiter((v for s in seqs for v in s), dtype=np.int32, count=int(off[-1])) if off[-1] else np.zeros(1, np.int32)
    return flat, off


def edit_distances(preds: Sequence[Sequence[int]], refs: Sequence[Sequence[int]], device="cuda") -> torch.Tensor:
    """int32 [n] Levenshtein distances of n symbol-sequence pairs (device tensor)."""
    if len(preds) != len(refs):
        raise ValueError("edit_distances: preds and refs must have the same number of sequences")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("edit_distances: GPU only (no CPU fallback)")
    a, a_off = _flatten(preds)
    b, b_off = _flatten(refs)
    host = torch.from_numpy(np.concatenate([a, a_off, b, b_off])).pin_memory()
    d = host.to(dev, non_blocking=True)
    pa, po, pb, pbo = d[: a.size], d[a.size: a.size + a_off.size], d[a.size + a_off.size: a.size + a_off.size + b.size], d[-b_off.size:]
    dist = torch.empty(len(preds), dtype=torch.int32, device=dev)
    st = _lib.lib().ts_edit_distance(pa.data_ptr(), po.data_ptr(), pb.data_ptr(), pbo.data_ptr(), len(preds),
                                     int(max((len(s) for s in preds), default=0)), dist.data_ptr(),
                                     torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(st, "ts_edit_distance")
    return dist


class _ErrorRate:
    """Minimal torchmetrics.Metric look-alike: call / update with (preds, targets) string lists, compute(), reset()."""

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self.reset()

    def reset(self):
        self.errors = None                    # device scalar, created by the first update (the constructor must not touch the GPU)
        self.total = 0

    def _symbols(self, texts: List[str]):
        raise NotImplementedError

    def update(self, preds: Union[str, List[str]], target: Union[str, List[str]]):
        preds = [preds] if isinstance(preds, str) else list(preds)
        target = [target] if isinstance(target, str) else list(target)
        p, r = self._symbols(preds, target)
        e = edit_distances(p, r, self.device).sum().to(torch.float64)
        n = sum(len(s) for s in r)
        self.errors = e if self.errors is None else self.errors + e
        self.total += n
        return e, n

    def compute(self) -> torch.Tensor:
        if self.errors is None:
            return torch.zeros((), dtype=torch.float32)
        return (self.errors / max(self.total, 1)).to(torch.float32)

    def __call__(self, preds, target) -> torch.Tensor:
        """torchmetrics' `forward`: accumulate AND return the value of THIS batch (`compute()` gives the running value; the reference's
        validation_step logs the metric object, which Lightning resolves with compute() at epoch end, module.py:153-163)."""
        e, n = self.update(preds, target)
        return (e / max(n, 1)).to(torch.float32)

    def to(self, device):
        self.device = torch.device(device)
        if self.errors is not None:
            self.errors = self.errors.to(self.device)
        return self


class CharErrorRate(_ErrorRate):
    def _symbols(self, preds, refs):
        return [[ord(c) for c in s] for s in preds], [[ord(c) for c in s] for s in refs]


class WordErrorRate(_ErrorRate):
    def _symbols(self, preds, refs):
        ids = {}
        enc = lambda s: [ids.setdefault(w, len(ids)) for w in s.split()]
        return [enc(s) for s in preds], [enc(s) for s in refs]
