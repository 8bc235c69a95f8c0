"""Activation containers at the Python boundary.

Internal activations are bf16 [B, C, pitch] buffers (time contiguous, pitch = ts_time_pitch(T), see
include/thunder_speech_amd.h).  They travel between modules as ordinary torch views `buf[:, :, :T]`, so the
reference's `(Tensor[B, C, T], lengths) -> (Tensor, lengths)` convention (blocks.py:94-115) is kept and a
chain of blocks never copies; a plain fp32 [B, C, T] tensor (reference layout) is accepted anywhere and
packed on entry.

Tail-zero invariant.  Buffers handed out by `arena()` are zero-initialised ONCE, carry TS_GUARD_BYTES of zeros on
both sides, and every kernel that writes into them stores 0 for frames >= the clip's length.  A view produced that
way is tagged `_ts_tz = True`; launches whose inputs are all tagged run the mask-free fast kernels
(TS_TCS_IN_TAILZERO).  Arena buffers are owned by the library and REUSED by later calls with the same shape --
tensors the caller keeps (encoder output, logits) are therefore always freshly allocated instead.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import os

import torch

from . import _lib

_ARENA: Dict[Tuple, torch.Tensor] = {}
_GUARD = _lib.GUARD_BYTES // 2          # elements (bf16)


def require_gpu(x: torch.Tensor, what: str) -> None:
    if not x.is_cuda:
        raise RuntimeError(
            f"{what}: thunder_speech_amd computes on the MI355X only (got a {x.device} tensor); there is no CPU "
            "fallback -- move the module and its inputs to 'cuda'.")


def is_internal(x: torch.Tensor) -> bool:
    return (x.dim() == 3 and x.dtype == torch.bfloat16 and x.is_cuda and x.stride(2) == 1 and x.stride(1) % 8 == 0
            and x.stride(1) >= x.shape[2] and x.stride(0) == x.shape[1] * x.stride(1) and x.data_ptr() % 16 == 0)


def is_tail_zero(x: torch.Tensor) -> bool:
    return bool(getattr(x, "_ts_tz", False)) and x.stride(1) >= _lib.time_pitch(x.shape[2])


def tag_tail_zero(x: torch.Tensor) -> torch.Tensor:
    x._ts_tz = True
    return x


def backing(x: torch.Tensor) -> torch.Tensor:
    """The full [B, C, pitch] buffer behind an internal view."""
    b, c, _ = x.shape
    return x.as_strided((b, c, x.stride(1)), (x.stride(0), x.stride(1), 1))


def alloc(b: int, c: int, t: int, device, dtype=torch.bfloat16) -> torch.Tensor:
    """Fresh (caller-owned) buffer; contents undefined."""
    return torch.empty(b, c, _lib.time_pitch(t), device=device, dtype=dtype)


def arena(slot, b: int, c: int, t: int, device) -> torch.Tensor:
    """Library-owned, zero-initialised, guarded bf16 buffer [B, C, pitch] for (slot, shape, device, STREAM); reused across calls.
    `slot` names the owner (callers put id(module) into it), so two modules of equal shape never share scratch, and the
    current stream is part of the key, so the same module driven from two streams (or eagerly and under graph capture) gets
    separate buffers instead of racing on one."""
    pitch = _lib.time_pitch(t)
    dev = torch.device(device)
    key = (slot, b, c, pitch, str(dev), torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
    flat = _ARENA.get(key)
    if flat is None:
        flat = torch.zeros(b * c * pitch + 2 * _GUARD, dtype=torch.bfloat16, device=device)
        _ARENA[key] = flat
    return flat[_GUARD: _GUARD + b * c * pitch].view(b, c, pitch)


def release_arena() -> None:
    _ARENA.clear()


def pack(x: torch.Tensor, lengths: Optional[torch.Tensor] = None, slot="pack") -> torch.Tensor:
    """fp32/any [B, C, T] (reference layout) -> internal bf16 view [B, C, T].  With `lengths` the frames >= length are
    zeroed and the result satisfies the tail-zero invariant (it then lives in the arena)."""
    require_gpu(x, "pack")
    if is_internal(x):
        return x
    b, c, t = x.shape
    src = x.to(torch.float32).contiguous()
    li = lengths_i32(lengths, x.device) if lengths is not None else None
    buf = arena((slot, "in"), b, c, t, x.device) if li is not None else alloc(b, c, t, x.device)
    st = _lib.lib().ts_pack_activation(src.data_ptr(), li.data_ptr() if li is not None else None, b, c, t,
                                       buf.data_ptr(), buf.shape[2], torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(st, "ts_pack_activation")
    out = buf[:, :, :t]
    return tag_tail_zero(out) if li is not None else out


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


_LEN_SCOPE = None


class lengths_scope:
    """One forward pass of an encoder: inside it, lengths_i32 converts a given lengths tensor object once and hands the result to every
    later caller (the same object flows through all length-preserving blocks).  Unlike the attribute cache below, this also holds while
    a hipGraph is being captured -- the conversion is recorded once per forward instead of once per block (37 tiny copy kernels in a
    QuartzNet15x5 step otherwise, ~5 us each even inside a graph) -- and nothing outlives the forward, so no capture sees another's
    tensors."""

    def __enter__(self):
        global _LEN_SCOPE
        self._prev, _LEN_SCOPE = _LEN_SCOPE, (dict(_LEN_SCOPE) if _LEN_SCOPE else {})      # a nested scope sees the outer one's entries
        return self

    def __exit__(self, *exc):
        global _LEN_SCOPE
        _LEN_SCOPE = self._prev
        return False


_FULL = {}


def full_lengths(b: int, t: int, device) -> torch.Tensor:
    """int32 [b] filled with t ("every frame is valid": the decoder convs are not masked).  Constant, so it is made once per (b, t,
    device) instead of once per forward -- except while a hipGraph is being captured, where a cached tensor must not be born inside
    the graph's private pool."""
    key = (b, t, str(torch.device(device)))
    hit = _FULL.get(key)
    if hit is not None:
        return hit
    out = torch.full((b,), t, dtype=torch.int32, device=device)
    if not (out.is_cuda and torch.cuda.is_current_stream_capturing()):
        if len(_FULL) > 256:
            _FULL.clear()
        _FULL[key] = out
    return out


_KINDS = {torch.float32: 0, torch.int64: 1, torch.int32: 2}
_NO_MAP = False        # tools may set this to time the plain torch expressions instead


def lengths_map(lengths: torch.Tensor, add: int, div: int, plus: int, out_dtype=None) -> torch.Tensor:
    """floor((lengths + add) / div) + plus in ONE launch (ts_lengths_map) for f32 / int64 / int32 device lengths, in the input type's
    arithmetic; the int32 copy the kernels need is produced by the same launch and remembered for lengths_i32 (inside a lengths_scope).
    Anything else (CPU tensors, other dtypes) takes the plain torch expression."""
    out_dtype = out_dtype or lengths.dtype
    if _NO_MAP or not (lengths.is_cuda and lengths.dim() == 1 and lengths.is_contiguous() and lengths.dtype in _KINDS and out_dtype in _KINDS):
        r = torch.div(lengths + add, div, rounding_mode="floor") + plus
        return r.to(out_dtype)
    out = torch.empty(lengths.shape, dtype=out_dtype, device=lengths.device)
    i32 = out if out_dtype == torch.int32 else torch.empty(lengths.shape, dtype=torch.int32, device=lengths.device)
    st = _lib.lib().ts_lengths_map(lengths.data_ptr(), _KINDS[lengths.dtype], out.data_ptr(), _KINDS[out_dtype],
                                   None if i32 is out else i32.data_ptr(), lengths.numel(), int(add), int(div), int(plus),
                                   torch.cuda.current_stream(lengths.device).cuda_stream)
    _lib.check(st, "ts_lengths_map")
    if _LEN_SCOPE is not None:
        _LEN_SCOPE[id(out)] = (out, out._version, i32)
    return out


def remember_i32(lengths: torch.Tensor, i32: torch.Tensor) -> None:
    """`i32` IS the int32 copy of `lengths` (written by the launch that produced both): lengths_i32 hands it out inside the current lengths_scope."""
    if _LEN_SCOPE is not None:
        _LEN_SCOPE[id(lengths)] = (lengths, lengths._version, i32)


def lengths_i32(lengths: torch.Tensor, device) -> torch.Tensor:
    """Reference lengths may be float or int (A5); the kernels take floor()ed int32 on the device."""
    if lengths.dtype == torch.int32 and lengths.device == torch.device(device) and lengths.is_contiguous():
        return lengths
    if _LEN_SCOPE is not None:
        hit = _LEN_SCOPE.get(id(lengths))
        if hit is not None and hit[0] is lengths and hit[1] == lengths._version and hit[2].device == torch.device(device):
            return hit[2]
        out = lengths.to(device=device, dtype=torch.int64).to(torch.int32).contiguous()
        _LEN_SCOPE[id(lengths)] = (lengths, lengths._version, out)
        return out
    # while a hipGraph is being captured the conversion must be RECORDED (a replay refreshes `lengths` in place and the
    # kernels have to see the new values), so the cache is neither read nor written
    capturing = lengths.is_cuda and torch.cuda.is_current_stream_capturing()
    cached = getattr(lengths, "_ts_i32", None)
    if not capturing and cached is not None and cached[0] == lengths._version and cached[1].device == torch.device(device):
        return cached[1]
    out = lengths.to(device=device, dtype=torch.int64).to(torch.int32).contiguous()
    if capturing:
        return out
    try:
        lengths._ts_i32 = (lengths._version, out)     # the same tensor object flows through the length-preserving blocks
    except Exception:
        pass
    return out
