"""Activation containers at the Python boundary.

Internal activations are bf16 [B, C, pitch] buffers (time contiguous, pitch = multiple of 128, see
include/thunder_speech_amd.h).  They travel between modules as ordinary torch views `buf[:, :, :T]`, so the
reference's `(Tensor[B, C, T], lengths) -> (Tensor, lengths)` convention (blocks.py:94-115) is kept and a
chain of blocks never copies; a plain fp32 [B, C, T] tensor (reference layout) is accepted anywhere and
packed on entry.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import torch

from . import _lib


def require_gpu(x: torch.Tensor, what: str) -> None:
    if not x.is_cuda:
        raise RuntimeError(
            f"{what}: thunder_speech_amd computes on the MI355X only (got a {x.device} tensor); there is no CPU "
            "fallback -- move the module and its inputs to 'cuda'.")


def is_internal(x: torch.Tensor) -> bool:
    return (x.dim() == 3 and x.dtype == torch.bfloat16 and x.is_cuda and x.stride(2) == 1 and x.stride(1) % 8 == 0
            and x.stride(1) >= x.shape[2] and x.stride(0) == x.shape[1] * x.stride(1) and x.data_ptr() % 16 == 0)


def backing(x: torch.Tensor) -> torch.Tensor:
    """The full [B, C, pitch] buffer behind an internal view."""
    b, c, _ = x.shape
    return x.as_strided((b, c, x.stride(1)), (x.stride(0), x.stride(1), 1))


def alloc(b: int, c: int, t: int, device, dtype=torch.bfloat16) -> torch.Tensor:
    return torch.empty(b, c, _lib.time_pitch(t), device=device, dtype=dtype)


def pack(x: torch.Tensor) -> torch.Tensor:
    """fp32/any [B, C, T] (reference layout) -> internal bf16 view [B, C, T]."""
    require_gpu(x, "pack")
    if is_internal(x):
        return x
    b, c, t = x.shape
    src = x.to(torch.float32).contiguous()
    buf = alloc(b, c, t, x.device)
    st = _lib.lib().ts_pack_activation(src.data_ptr(), b, c, t, buf.data_ptr(), buf.shape[2],
                                       torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(st, "ts_pack_activation")
    return buf[:, :, :t]


def unpack(x: torch.Tensor) -> torch.Tensor:
    """internal view -> contiguous fp32 [B, C, T]."""
    if not is_internal(x):
        return x
    b, c, t = x.shape
    out = torch.empty(b, c, t, device=x.device, dtype=torch.float32)
    st = _lib.lib().ts_unpack_activation(x.data_ptr(), b, c, t, x.stride(1), out.data_ptr(),
                                         torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(st, "ts_unpack_activation")
    return out


def lengths_i32(lengths: torch.Tensor, device) -> torch.Tensor:
    """Reference lengths may be float or int (A5); the kernels take floor()ed int32 on the device."""
    if lengths.dtype == torch.int32 and lengths.device == torch.device(device) and lengths.is_contiguous():
        return lengths
    cached = getattr(lengths, "_ts_i32", None)
    if cached is not None and cached[0] == lengths._version and cached[1].device == torch.device(device):
        return cached[1]
    out = lengths.to(device=device, dtype=torch.int64).to(torch.int32).contiguous()
    try:
        lengths._ts_i32 = (lengths._version, out)     # the same tensor object flows through the length-preserving blocks
    except Exception:
        pass
    return out
