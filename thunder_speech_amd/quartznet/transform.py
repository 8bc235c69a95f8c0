"""Mel-filterbank front end -- reference names and constructor signature of
src/thunder/quartznet/transform.py, executed by the fused HIP front end (csrc/frontend.hip).

`FilterbankFeatures(...)` returns a MultiSequential with the reference's four children (so
`filterbank[1].stft_func`, the "1.window" / "2.layer.0.fb" buffers and `patch_stft` keep working) whose
forward is ONE call of ts_mel_frontend_fwd.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import numpy as np
import torch
from torch import nn

from .. import _lib
from .. import rng as _rng
from .. import tensors as _t
from ..blocks import Masked, MultiSequential
from .spec_augment import SpecAugment, SpecCutout

__all__ = ["FeatureBatchNormalizer", "DitherAudio", "PreEmphasisFilter", "PowerSpectrum", "MelScale",
           "FilterbankFeatures", "patch_stft", "melscale_fbanks"]


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """Slaney-scale, slaney-normalised triangular filterbank [n_freqs, n_mels] -- what the reference obtains from
    torchaudio.functional.melscale_fbanks(norm="slaney", mel_scale="slaney") (transform.py:227-236)."""
    def hz_to_mel(f):
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) / (math.log(6.4) / 27.0), f * 3.0 / 200.0)

    def mel_to_hz(m):
        return np.where(m >= 15.0, 1000.0 * np.exp((math.log(6.4) / 27.0) * (m - 15.0)), m * 200.0 / 3.0)

    freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    pts = mel_to_hz(np.linspace(hz_to_mel(np.float64(f_min)), hz_to_mel(np.float64(f_max)), n_mels + 2))
    width = np.diff(pts)
    rel = pts[None, :] - freqs[:, None]
    fb = np.clip(np.minimum(-rel[:, :-2] / width[:-1], rel[:, 2:] / width[1:]), 0.0, None)
    fb *= (2.0 / (pts[2:] - pts[:-2]))[None, :]
    return torch.from_numpy(fb.astype(np.float32))


def _f32(x: torch.Tensor, what: str) -> torch.Tensor:
    _t.require_gpu(x, what)
    return x.to(torch.float32).contiguous()


def _stream(x: torch.Tensor):
    return torch.cuda.current_stream(x.device).cuda_stream


# The five stage modules also run ON THEIR OWN (the reference's tests call them directly with arbitrary parameters,
# tests/quartznet/test_transform_qn.py:130-260): generic single-stage kernels of csrc/frontend_stages.hip, reference layout, f32.
# Inside FilterbankFeatures none of these forwards is called -- the fused front end computes all stages in two launches.
class FeatureBatchNormalizer(nn.Module):
    def __init__(self):
        super().__init__()
        self.div_guard = 1e-5

    @torch.no_grad()
    def forward(self, x: torch.Tensor, lengths: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """normalize_tensor with the length mask (reference transform.py:77-92 -> blocks.py:136-149, quirk A1)."""
        xf = _f32(x, "FeatureBatchNormalizer")
        b, f, t = xf.shape
        out = torch.empty_like(xf)
        st = _lib.lib().ts_fe_normalize(xf.data_ptr(), _t.lengths_i32(lengths, xf.device).data_ptr(), out.data_ptr(), b, f, t,
                                        float(self.div_guard), _stream(xf))
        _lib.check(st, "ts_fe_normalize")
        return out, lengths


class DitherAudio(nn.Module):
    def __init__(self, dither: float = 1e-5):
        super().__init__()
        self.dither = dither

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x + dither * N(0, 1) in training mode, identity in eval mode (reference transform.py:109-118)."""
        if not self.training:
            return x
        xf = _f32(x, "DitherAudio")
        out = torch.empty_like(xf)
        st = _lib.lib().ts_fe_dither(xf.data_ptr(), out.data_ptr(), xf.shape[0], xf.shape[1], float(self.dither), _rng.next_seed(), _stream(xf))
        _lib.check(st, "ts_fe_dither")
        return out


class PreEmphasisFilter(nn.Module):
    def __init__(self, preemph: float = 0.97):
        super().__init__()
        self.preemph = preemph

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """y[0] = x[0], y[n] = x[n] - preemph * x[n-1] (reference transform.py:136-144)."""
        xf = _f32(x, "PreEmphasisFilter")
        out = torch.empty_like(xf)
        st = _lib.lib().ts_fe_preemph(xf.data_ptr(), out.data_ptr(), xf.shape[0], xf.shape[1], float(self.preemph), _stream(xf))
        _lib.check(st, "ts_fe_preemph")
        return out


class PowerSpectrum(nn.Module):
    def __init__(self, n_window_size: int = 320, n_window_stride: int = 160, n_fft: Optional[int] = None):
        super().__init__()
        if n_window_size <= 0 or n_window_stride <= 0:
            raise ValueError(f"{self} got an invalid value for either n_window_size or n_window_stride. "
                             "Both must be positive ints.")
        self.win_length = n_window_size
        self.hop_length = n_window_stride
        self.n_fft = n_fft or 2 ** math.ceil(math.log2(self.win_length))
        self.register_buffer("window", torch.hann_window(self.win_length, periodic=False))
        self.stft_func = torch.stft          # kept for patch_stft(); the HIP front end does not call it

    def get_sequence_length(self, lengths: torch.Tensor) -> torch.Tensor:
        if lengths.is_cuda and lengths.dtype in (torch.float32, torch.int64, torch.int32) and not _t._NO_MAP:
            return _t.lengths_map(lengths, 0, self.hop_length, 1, out_dtype=torch.long)      # one launch (+ the int32 copy for the kernels)
        return (torch.floor(lengths / self.hop_length) + 1).to(dtype=torch.long)

    def _stft_tables(self, device):
        key = (str(device), self.window.data_ptr(), self.window._version)
        if getattr(self, "_tab_key", None) != key:
            win = torch.zeros(self.n_fft, dtype=torch.float32)
            left = (self.n_fft - self.win_length) // 2
            win[left:left + self.win_length] = self.window.detach().float().cpu()
            ang = 2.0 * math.pi * torch.arange(self.n_fft, dtype=torch.float64) / self.n_fft
            tw = torch.stack([torch.cos(ang), torch.sin(ang)], dim=1).to(torch.float32)
            self._tab, self._tab_key = (win.to(device), tw.contiguous().to(device)), key
        return self._tab

    @torch.no_grad()
    def forward(self, x: torch.Tensor, lengths: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """|STFT|^2 [B, n_fft/2 + 1, T // hop + 1] and the frame counts (reference transform.py:186-208)."""
        xf = _f32(x, "PowerSpectrum")
        b, n = xf.shape
        if self.win_length > self.n_fft:
            raise RuntimeError(f"PowerSpectrum: win_length {self.win_length} > n_fft {self.n_fft} (torch.stft refuses it too)")
        win, tw = self._stft_tables(xf.device)
        frames = 1 + (n + 2 * (self.n_fft // 2) - self.n_fft) // self.hop_length      # torch.stft(center=True): n // hop + 1 for even n_fft
        out = torch.empty(b, self.n_fft // 2 + 1, frames, dtype=torch.float32, device=xf.device)
        st = _lib.lib().ts_fe_power_spectrum(xf.data_ptr(), win.data_ptr(), tw.data_ptr(), out.data_ptr(), b, n, self.n_fft, self.hop_length,
                                             _stream(xf))
        _lib.check(st, "ts_fe_power_spectrum")
        return out, self.get_sequence_length(lengths)


class MelScale(nn.Module):
    def __init__(self, sample_rate: int, n_fft: int, nfilt: int, log_scale: bool = True):
        super().__init__()
        fb = melscale_fbanks(int(1 + n_fft // 2), 0.0, sample_rate / 2, nfilt, sample_rate).transpose(0, 1).unsqueeze(0)
        self.register_buffer("fb", fb.contiguous())
        self.log_scale = log_scale

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """log(fb @ x + 2^-24) (or the plain product), [B, n_freq, T] -> [B, nfilt, T] (reference transform.py:243-255)."""
        xf = _f32(x, "MelScale")
        b, f, t = xf.shape
        fb = self.fb.to(device=xf.device, dtype=torch.float32).contiguous()
        if fb.shape[2] != f:
            raise RuntimeError(f"MelScale: the input has {f} frequency bins, the filterbank {fb.shape[2]}")
        out = torch.empty(b, fb.shape[1], t, dtype=torch.float32, device=xf.device)
        st = _lib.lib().ts_fe_mel(xf.data_ptr(), fb.data_ptr(), out.data_ptr(), b, f, fb.shape[1], t, int(bool(self.log_scale)), _stream(xf))
        _lib.check(st, "ts_fe_mel")
        return out


class _FilterbankFeatures(MultiSequential):
    """MultiSequential(Masked(Dither, PreEmphasis), PowerSpectrum, Masked(MelScale), FeatureBatchNormalizer)."""

    def _tables(self, device):
        ps, mel = self[1], self[2].layer[0]
        key = (str(device), ps.window.data_ptr(), ps.window._version, mel.fb.data_ptr(), mel.fb._version)
        if getattr(self, "_tab_key", None) != key:
            win = torch.zeros(ps.n_fft, dtype=torch.float32)
            left = (ps.n_fft - ps.win_length) // 2
            win[left:left + ps.win_length] = ps.window.detach().float().cpu()
            fb = mel.fb.detach().float().cpu()[0]                       # [n_mels, n_freqs]
            weights, offsets = [], []
            for m in range(fb.shape[0]):
                nz = torch.nonzero(fb[m]).flatten()
                first = int(nz[0]) if len(nz) else 0
                last = int(nz[-1]) + 1 if len(nz) else 0
                offsets.append((first, len(weights)))
                weights.extend(fb[m, first:last].tolist())
            offsets.append((0, len(weights)))
            self._tab = (win.to(device), torch.tensor(weights, dtype=torch.float32).to(device),
                         torch.tensor(offsets, dtype=torch.int32).to(device), len(weights))
            self._tab_key = key
        return self._tab

    def forward(self, audio: torch.Tensor, audio_lengths: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        _t.require_gpu(audio, "FilterbankFeatures")
        # DitherAudio is a training-only step (transform.py:115); the augmentation modules follow their own training flag
        dither = float(self[0].layer[0].dither) if self[0].layer[0].training else 0.0
        ps, mel = self[1], self[2].layer[0]
        if not mel.log_scale:
            # MelScale(log_scale=False) (reference transform.py:213, :253: the plain filterbank product, no log): no model of the reference is
            # built this way and the fused kernel has the log inside its spectrum loop, so this configuration runs the reference's module
            # chain stage by stage (MultiSequential.forward), every stage on its own HIP kernel (ts_fe_*)
            return super().forward(audio, audio_lengths)
        x = audio.to(torch.float32).contiguous()
        b, n = x.shape
        win, mw, moff, nnz = self._tables(x.device)
        d = _lib.FrontendDesc()
        d.batch, d.n_samples, d.n_fft, d.hop, d.win_length = b, n, ps.n_fft, ps.hop_length, ps.win_length
        d.n_mels = mel.fb.shape[1]
        d.preemph = float(self[0].layer[1].preemph)
        d.n_frames = n // ps.hop_length + 1
        d.pitch_out = _lib.time_pitch(d.n_frames)
        d.window, d.mel_weights, d.mel_offsets, d.mel_nnz = win.data_ptr(), mw.data_ptr(), moff.data_ptr(), nnz
        if dither > 0:
            # DitherAudio (transform.py:109-118): x + dither * N(0, 1), drawn inside the kernel's sample load from a Philox
            # stream keyed by (seed, clip, sample): no noise tensor, no extra pass over the waveform
            d.dither, d.dither_seed = dither, _rng.next_seed()
        tables = [m.layer[0].draw(d.n_mels, d.n_frames, x.device) for m in list(self)[4:] if m.layer[0].training]
        tables = [t for t in tables if t is not None]
        if tables:
            # SpecCutout / SpecAugment (transform.py:299-320): the rectangles are zeroed by the normaliser as it writes the features
            table = tables[0] if len(tables) == 1 else torch.cat(tables)
            d.masks, d.n_masks = table.data_ptr(), table.shape[0]
        L = _lib.lib()
        ws_bytes = L.ts_frontend_workspace_bytes(C.byref(d))
        if ws_bytes < 0:
            _lib.check(int(ws_bytes), "ts_frontend_workspace_bytes")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        feats = _t.arena(("feat", id(self)), b, d.n_mels, d.n_frames, x.device)   # normalize_kernel zeroes frames >= length up to the pitch
        flen = torch.empty(b, dtype=torch.int32, device=x.device)
        wl = _t.lengths_i32(audio_lengths, x.device)
        # the lengths this forward returns (PowerSpectrum.get_sequence_length: floor(len / hop) + 1 as int64) come out of the same launch; the
        # kernel's arithmetic -- floor(len) to int32 first, then the integer division -- equals the reference's float expression for len >= 0
        fused_len = audio_lengths.is_cuda and audio_lengths.dtype in (torch.float32, torch.int64, torch.int32) and not _t._NO_MAP
        flen64 = torch.empty(b, dtype=torch.int64, device=x.device) if fused_len else None
        if flen64 is not None:
            d.feat_len64 = flen64.data_ptr()
        st = L.ts_mel_frontend_fwd(C.byref(d), x.data_ptr(), wl.data_ptr(), feats.data_ptr(), flen.data_ptr(),
                                   ws.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(st, "ts_mel_frontend_fwd")
        self._last_logmel = ws[: b * d.n_frames * d.n_mels * 4].view(torch.float32).view(b, d.n_frames, d.n_mels)
        if flen64 is None:
            return _t.tag_tail_zero(feats[:, :, :d.n_frames]), ps.get_sequence_length(audio_lengths)
        _t.remember_i32(flen64, flen)        # the encoder's kernels take the int32 copy the same launch wrote
        return _t.tag_tail_zero(feats[:, :, :d.n_frames]), flen64

    def last_logmel(self) -> torch.Tensor:
        """Parity hook: un-normalised log-mel [B, frames, n_mels] of the last forward (kernel-1 output)."""
        return self._last_logmel


def FilterbankFeatures(sample_rate: int = 16000, n_window_size: int = 320, n_window_stride: int = 160, n_fft: int = 512,
                       preemph: float = 0.97, nfilt: int = 64, dither: float = 1e-5, num_cutout_masks: int = 0,
                       num_time_masks: int = 0, num_freq_masks: int = 0, mask_time_width: int = 50,
                       mask_freq_width: int = 20) -> nn.Module:
    """Same signature as the reference (quartznet/transform.py:258-271)."""
    if num_cutout_masks > 0 and (num_freq_masks + num_time_masks > 0):
        raise ValueError("Cutout and SpecAugment can't be used at the same time.")
    modules = [
        Masked(DitherAudio(dither=dither), PreEmphasisFilter(preemph=preemph)),
        PowerSpectrum(n_window_size=n_window_size, n_window_stride=n_window_stride, n_fft=n_fft),
        Masked(MelScale(sample_rate=sample_rate, n_fft=n_fft, nfilt=nfilt)),
        FeatureBatchNormalizer(),
    ]
    if num_cutout_masks > 0:
        modules.append(Masked(SpecCutout(rect_masks=num_cutout_masks, time_width=mask_time_width, freq_width=mask_freq_width)))
    if num_freq_masks + num_time_masks > 0:
        modules.append(Masked(SpecAugment(time_masks=num_time_masks, freq_masks=num_freq_masks, time_width=mask_time_width,
                                          freq_width=mask_freq_width)))
    return _FilterbankFeatures(*modules)


def patch_stft(filterbank: nn.Module) -> nn.Module:
    """Reference hook for FFT-less export targets (transform.py:324-336): `filterbank[1].stft_func = convolution_stft`.  The fused HIP
    front end has its own FFT and never calls `stft_func`, so inside FilterbankFeatures this only records the request, as in the
    reference's attribute; `blocks.convolution_stft` itself is callable (one launch of the direct-DFT stage kernel)."""
    from ..blocks import convolution_stft
    filterbank[1].stft_func = convolution_stft
    return filterbank
