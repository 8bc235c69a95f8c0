"""TEST INFRASTRUCTURE (CPU oracle) -- validation metrics of reference module.py:153-156.

torchmetrics (pinned ^0.9 by the reference, absent from this image) defines CharErrorRate = sum_i edit(pred_i, ref_i) /
sum_i len(ref_i) over characters and WordErrorRate the same over whitespace-split words, with the unit-cost Levenshtein
distance.  Plain-Python dynamic programme (small cases only)."""
from __future__ import annotations

from typing import List, Sequence


def edit_distance(a: Sequence, b: Sequence) -> int:
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, y in enumerate(b, 1):
            cur[j] = min(prev[j - 1] + (x != y), prev[j] + 1, cur[j - 1] + 1)
        prev = cur
    return prev[len(b)]


def char_error_rate(preds: List[str], refs: List[str]) -> float:
    return sum(edit_distance(list(p), list(r)) for p, r in zip(preds, refs)) / max(sum(len(r) for r in refs), 1)


def word_error_rate(preds: List[str], refs: List[str]) -> float:
    return sum(edit_distance(p.split(), r.split()) for p, r in zip(preds, refs)) / max(sum(len(r.split()) for r in refs), 1)
