"""Shared building blocks -- mirror of the reference's src/thunder/blocks.py public names.

`MultiSequential`, `Masked`, `lengths_to_mask`, `get_same_padding`, `conv1d_decoder`, `linear_decoder`,
`SwapLastDimension`, `normalize_tensor`, `convolution_stft` keep the reference signatures (blocks.py:9-19).  The decoders run
the fused pointwise HIP kernel; the small tensor helpers are host-side plumbing.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from . import _lib
from . import plan as _plan
from . import tensors as _t

__all__ = ["convolution_stft", "MultiSequential", "Masked", "normalize_tensor", "lengths_to_mask", "get_same_padding",
           "conv1d_decoder", "SwapLastDimension", "linear_decoder"]


def _fourier_matrix(n_fft: int, device) -> Tensor:
    """The DFT matrix exp(-2 pi i j k / n) as complex64 [n, n] (reference blocks.py:29-35).  The phase j k is reduced mod n in integer
    arithmetic before the trigonometry, so large n stay exact to f32 (the reference's f32 outer product drifts; its test allows 1e-3)."""
    idx = torch.arange(n_fft, dtype=torch.int64)
    ang = -2.0 * math.pi * ((idx[:, None] * idx[None, :]) % n_fft).to(torch.float64) / n_fft
    return torch.complex(torch.cos(ang), torch.sin(ang)).to(torch.complex64).to(device)


def convolution_stft(input_data: Tensor, n_fft: int = 1024, hop_length: int = 512, win_length: int = 1024,
                     window: Optional[Tensor] = None, center: bool = True, return_complex: bool = False) -> Tensor:
    """The reference's export-friendly STFT (blocks.py:38-91; `patch_stft` swaps it in for torch.stft): reflect-pad by n_fft / 2, frames of
    n_fft samples every hop_length, the win_length window centred in zeros, one-sided transform; returns [B, n_fft/2+1, frames, 2] =
    (real, imaginary) whatever `center` / `return_complex` say, as the reference does.  Here: one launch of the direct-DFT stage kernel
    (ts_fe_stft), GPU tensors only."""
    assert n_fft >= win_length
    _t.require_gpu(input_data, "convolution_stft")
    if window is None:
        window = torch.hann_window(win_length, periodic=False)
    x = input_data.detach().reshape(input_data.shape[0], -1).to(torch.float32).contiguous()
    b, n = x.shape
    if n_fft // 2 >= n:
        raise RuntimeError(f"convolution_stft: reflect padding of {n_fft // 2} needs more than {n} samples")
    win = torch.zeros(n_fft, dtype=torch.float32)
    left = (n_fft - win_length) // 2
    win[left:left + win_length] = window.detach().float().cpu()
    ang = 2.0 * math.pi * torch.arange(n_fft, dtype=torch.float64) / n_fft
    tw = torch.stack([torch.cos(ang), torch.sin(ang)], dim=1).to(torch.float32).contiguous()
    frames = 1 + (n + 2 * (n_fft // 2) - n_fft) // hop_length          # torch.stft(center=True); == n // hop + 1 for even n_fft
    out = torch.empty(b, n_fft // 2 + 1, frames, 2, dtype=torch.float32, device=x.device)
    win_d, tw_d = win.to(x.device), tw.to(x.device)          # named: they must outlive the launch
    st = _lib.lib().ts_fe_stft(x.data_ptr(), win_d.data_ptr(), tw_d.data_ptr(), out.data_ptr(), b, n, n_fft, hop_length,
                               torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(st, "ts_fe_stft")
    return out


class MultiSequential(nn.Sequential):
    """nn.Sequential with two inputs / outputs (reference blocks.py:94-102)."""

    def forward(self, audio: Tensor, audio_lengths: Tensor) -> Tuple[Tensor, Tensor]:
        for module in self.children():
            audio, audio_lengths = module(audio, audio_lengths)
        return audio, audio_lengths


class Masked(nn.Module):
    """Wraps single-input layers into the (x, lengths) convention (reference blocks.py:105-115).
    Kept for state-dict key parity ("...layer.0.weight"); fused parents never call it on the hot path."""

    def __init__(self, *layers):
        super().__init__()
        self.layer = nn.Sequential(*layers)

    def forward(self, audio: Tensor, audio_lengths: Tensor) -> Tuple[Tensor, Tensor]:
        return self.layer(audio), audio_lengths


def lengths_to_mask(lengths: Tensor, max_length: int) -> Tensor:
    """reference blocks.py:156-170"""
    lengths = lengths.type(torch.long)
    return torch.arange(max_length, device=lengths.device).expand(lengths.shape[0], max_length) < lengths.unsqueeze(1)


def get_same_padding(kernel_size: int, stride: int, dilation: int) -> int:
    """reference blocks.py:173-196"""
    if stride > 1 and dilation > 1:
        raise ValueError("Only stride OR dilation may be greater than 1")
    if dilation > 1:
        return (dilation * (kernel_size - 1) + 1) // 2
    return kernel_size // 2


def normalize_tensor(input_values: Tensor, mask: Optional[Tensor] = None, div_guard: float = 1e-7, dim: int = -1) -> Tensor:
    """reference blocks.py:118-153 (incl. quirk A1 for the masked branch).  Host-side helper used by the
    wav2vec2 pre-processing; the mel front end normalises inside its HIP kernel."""
    if mask is not None:
        valid = mask.type(torch.bool)
        x = torch.where(valid, input_values, torch.zeros_like(input_values))
        n = valid.sum(dim=dim, keepdim=True)
        mean = x.sum(dim=dim, keepdim=True) / n
        std = ((x - mean).pow(2).sum(dim=dim, keepdim=True) / n).sqrt()
        return torch.where(valid, (x - mean) / (std + div_guard), torch.zeros_like(x))
    mean = input_values.mean(dim=dim, keepdim=True)
    std = (input_values.var(dim=dim, keepdim=True) + div_guard).sqrt()
    return (input_values - mean) / std


class _PackedCache:
    """Lazily (re)builds packed kernel parameters when the fp32 master tensors change."""

    def __init__(self):
        self._key = None
        self._value = None

    def get(self, tensors, build):
        key = tuple((t.data_ptr(), t._version, str(t.device)) for t in tensors)
        if key != self._key:
            self._value = build()
            self._key = key
        return self._value


class _Conv1dDecoder(nn.Conv1d):
    """1x1 conv + bias -> fp32 logits [B, V, T'] (reference conv1d_decoder, blocks.py:199-216).
    state-dict keys: weight [V, C, 1], bias [V]."""

    def __init__(self, in_channels: int, num_classes: int):
        super().__init__(in_channels, num_classes, kernel_size=1, bias=True)
        self._cache = _PackedCache()

    def _build(self, pack_on_device: bool):
        return _plan.make_tcs_layer(self.weight.device, dw_w=None, pw_w=self.weight.detach(), bn=None, kernel=1, stride=1,
                                    dilation=1, padding=0, relu=False, bias_extra=self.bias.detach(), out_fp32=True,
                                    pack_on_device=pack_on_device)

    def _layer(self):
        return self._cache.get([self.weight, self.bias], lambda: self._build(False))

    def forward(self, x: Tensor) -> Tensor:
        _t.require_gpu(x, "conv1d_decoder")
        grad_on = torch.is_grad_enabled()
        if grad_on and x.requires_grad:
            # the encoder is being trained (whether or not the decoder's own parameters are frozen): fp32 training ops all the
            # way (GEMM forward / backward in train_ops.PointwiseConv); the bias add on the [B, V, T] logits is the one
            # broadcast left to autograd
            from .train_ops import PointwiseConv
            return PointwiseConv.apply(x, self.weight, True) + self.bias.view(1, -1, 1)       # f32 logits in either activation mode
        xi = _t.pack(x)
        b, _, t = xi.shape
        if self.training and grad_on and (self.weight.requires_grad or self.bias.requires_grad):
            return _DecoderFunction.apply(self.weight, self.bias, self, _t.backing(xi), t)
        full = _t.full_lengths(b, t, xi.device)   # the decoder conv is not masked
        y, t_out = self._layer().run(_t.backing(xi), t, full)
        return y[:, :, :t_out]


class _DecoderFunction(torch.autograd.Function):
    """Trainable 1x1 decoder over a frozen encoder: forward = the fused pointwise kernel (weights packed on the device,
    they change every step), backward = ts_decoder_bwd (dW, db)."""

    @staticmethod
    def forward(ctx, weight, bias, module, xb, t):
        b = xb.shape[0]
        full = _t.full_lengths(b, t, xb.device)
        y, t_out = module._build(True).run(xb, t, full)
        ctx.save_for_backward(xb)
        ctx.t, ctx.shape = t, weight.shape
        return y[:, :, :t_out]

    @staticmethod
    def backward(ctx, grad_logits):
        from . import _lib
        (xb,) = ctx.saved_tensors
        g = grad_logits.to(torch.float32).contiguous()
        b, v, t = g.shape
        c = xb.shape[1]
        dw = torch.empty(v, c, dtype=torch.float32, device=g.device)
        db = torch.empty(v, dtype=torch.float32, device=g.device)
        st = _lib.lib().ts_decoder_bwd(g.data_ptr(), xb.data_ptr(), b, v, c, t, g.stride(1), xb.stride(1), dw.data_ptr(),
                                       db.data_ptr(), torch.cuda.current_stream(g.device).cuda_stream)
        _lib.check(st, "ts_decoder_bwd")
        return dw.view(ctx.shape), db, None, None, None


def conv1d_decoder(decoder_input_channels: int, num_classes: int) -> nn.Module:
    decoder = _Conv1dDecoder(decoder_input_channels, num_classes)
    nn.init.xavier_uniform_(decoder.weight, gain=1.0)
    return decoder


class SwapLastDimension(nn.Module):
    def forward(self, x: Tensor) -> Tensor:
        return x.transpose(-1, -2)


class _LinearDecoder(nn.Sequential):
    """transpose -> dropout -> Linear -> transpose (reference linear_decoder, blocks.py:226-248); keys
    "2.weight" / "2.bias".  Eval mode runs the same fused pointwise kernel on the [B, C, T] layout (the two
    transposes cancel)."""

    def __init__(self, in_channels: int, num_classes: int, decoder_dropout: float):
        super().__init__(SwapLastDimension(), nn.Dropout(decoder_dropout), nn.Linear(in_channels, num_classes),
                         SwapLastDimension())
        self._cache = _PackedCache()

    def forward(self, x: Tensor) -> Tensor:
        _t.require_gpu(x, "linear_decoder")
        lin = self[2]
        drop = self[1]
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or lin.weight.requires_grad or lin.bias.requires_grad)
        if needs_grad or (drop.training and drop.p > 0):
            # training path (blocks.py:226-248: transpose -> dropout -> Linear -> transpose): dropout is elementwise and the
            # Linear over the channel axis is a 1x1 conv on [B, C, T], so the two transposes cancel here as well
            from . import train_ops as T
            h = T.dropout(T.to_act(x), drop.p, drop.training)
            return T.PointwiseConv.apply(h, lin.weight.unsqueeze(-1), True) + lin.bias.view(1, -1, 1)
        layer = self._cache.get([lin.weight, lin.bias], lambda: _plan.make_tcs_layer(
            lin.weight.device, dw_w=None, pw_w=lin.weight.detach(), bn=None, kernel=1, stride=1, dilation=1,
            padding=0, relu=False, bias_extra=lin.bias.detach(), out_fp32=True))
        xi = _t.pack(x)
        b, _, t = xi.shape
        full = _t.full_lengths(b, t, xi.device)
        y, t_out = layer.run(_t.backing(xi), t, full)
        return y[:, :, :t_out]


def linear_decoder(decoder_input_channels: int, num_classes: int, decoder_dropout: float) -> nn.Module:
    return _LinearDecoder(decoder_input_channels, num_classes, decoder_dropout)
