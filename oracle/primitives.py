"""Oracle restatement of the shared primitives (reference: src/thunder/blocks.py).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

import numpy as np
import torch


def lengths_to_mask(lengths: torch.Tensor, max_length: int) -> torch.Tensor:
    """mask[b, t] = t < floor(lengths[b])  (reference: blocks.py:156-170; lengths are cast to
    long first, so float lengths truncate)."""
    n = lengths.to(torch.long).reshape(-1, 1)
    t = torch.arange(max_length, device=lengths.device).reshape(1, -1)
    return t < n


def same_padding(kernel_size: int, stride: int, dilation: int) -> int:
    """reference: blocks.py:173-196.  K//2, or (d*(K-1)+1)//2 when dilated; stride and dilation
    may not both exceed 1."""
    if stride > 1 and dilation > 1:
        raise ValueError("Only stride OR dilation may be greater than 1")
    return (dilation * (kernel_size - 1) + 1) // 2 if dilation > 1 else kernel_size // 2


def conv_out_length(lengths: torch.Tensor, kernel_size: int, stride: int, padding: int, dilation: int) -> torch.Tensor:
    """reference: quartznet/blocks.py:142-156 (floor division that keeps the dtype of `lengths`)."""
    return torch.div(lengths + 2 * padding - dilation * (kernel_size - 1) - 1, stride, rounding_mode="floor") + 1


def masked_normalize(x: torch.Tensor, mask: torch.Tensor, div_guard: float) -> torch.Tensor:
    """Masked per-row normalisation over the last dim (reference: blocks.py:136-149).

    Quirk A1: the variance numerator is summed over ALL positions of the
    zero-filled tensor, so every padded position contributes mean**2; the divisor is the number of
    valid positions."""
    valid = mask.to(torch.bool)
    xz = torch.where(valid, x, torch.zeros_like(x))
    n = valid.sum(dim=-1, keepdim=True)
    mean = xz.sum(dim=-1, keepdim=True) / n
    centred_sq = (xz - mean) ** 2            # padded positions give mean**2 here
    std = (centred_sq.sum(dim=-1, keepdim=True) / n).sqrt()
    out = (xz - mean) / (std + div_guard)
    return torch.where(valid, out, torch.zeros_like(out))


def unmasked_normalize(x: torch.Tensor, div_guard: float) -> torch.Tensor:
    """reference: blocks.py:151-153 -- unbiased variance, guard added INSIDE the sqrt."""
    mean = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, keepdim=True)        # unbiased (N-1)
    return (x - mean) / (var + div_guard).sqrt()


def wav2vec2_preprocess(x: torch.Tensor, lengths: torch.Tensor, mask_input: bool, div_guard: float = 1e-7):
    """reference: huggingface/transform.py:34-55."""
    if mask_input:
        m = lengths_to_mask(lengths, x.shape[-1])
        return masked_normalize(x, m, div_guard), lengths
    return unmasked_normalize(x, div_guard), lengths


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bfloat16 and back (the HIP path's storage rounding)."""
    return x.to(torch.bfloat16).to(torch.float32)


def np_bf16_round(a: np.ndarray) -> np.ndarray:
    return bf16_round(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))).numpy()
