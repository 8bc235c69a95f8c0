"""Oracle restatement of the mel-filterbank front end.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: src/thunder/quartznet/transform.py
  DitherAudio            :95-118   (eval: identity)
  PreEmphasisFilter      :121-144
  PowerSpectrum          :147-208  (torch.stft centre/reflect, symmetric hann zero-padded to n_fft)
  MelScale               :211-255  (torchaudio.functional.melscale_fbanks slaney/slaney, log(x + 2^-24))
  FeatureBatchNormalizer :71-92    -> blocks.py:136-149 (masked normalise, quirk A1)

Third-party arithmetic restated here: `torchaudio.functional.melscale_fbanks` (torchaudio 0.12.0 is
pinned in the reference's poetry.lock and is absent from /root/reference and from this image).  The
restatement follows the public Slaney / librosa formula and is cross-checked in
tests/test_oracle_frontend.py against `transformers.audio_utils.mel_filter_bank` (independent
implementation).  The reference's own tests pin only its shape / finiteness, so the filterbank
VALUES are "parity unpinned" by reference tests; everything else here is pinned by fixtures produced
from the real reference modules.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

from .primitives import lengths_to_mask, masked_normalize

LOG_FLOOR = 2.0 ** -24


@dataclass(frozen=True)
class FrontendConfig:
    """Constructor arguments of reference FilterbankFeatures (transform.py:258-271)."""
    sample_rate: int = 16000
    n_window_size: int = 320
    n_window_stride: int = 160
    n_fft: int = 512
    preemph: float = 0.97
    nfilt: int = 64
    dither: float = 1e-5

    @property
    def n_freqs(self) -> int:
        return self.n_fft // 2 + 1


def hz_to_mel_slaney(f: np.ndarray) -> np.ndarray:
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3.0)
    logstep = math.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log = 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) / logstep
    return np.where(f >= 1000.0, log, lin)


def mel_to_hz_slaney(m: np.ndarray) -> np.ndarray:
    m = np.asarray(m, dtype=np.float64)
    lin = m * (200.0 / 3.0)
    logstep = math.log(6.4) / 27.0
    log = 1000.0 * np.exp(logstep * (m - 15.0))
    return np.where(m >= 15.0, log, lin)


def slaney_mel_filterbank(n_freqs: int, n_mels: int, sample_rate: int,
                          f_min: float = 0.0, f_max: float | None = None) -> np.ndarray:
    """Triangular slaney-scale, slaney-normalised filterbank, shape [n_mels, n_freqs] float32.

    Restates torchaudio.functional.melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate,
    norm="slaney", mel_scale="slaney") transposed (call site transform.py:227-239)."""
    f_max = float(sample_rate) / 2 if f_max is None else float(f_max)
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    m_pts = np.linspace(hz_to_mel_slaney(f_min), hz_to_mel_slaney(f_max), n_mels + 2)
    f_pts = mel_to_hz_slaney(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]                       # [n_mels+1]
    slopes = f_pts[None, :] - all_freqs[:, None]          # [n_freqs, n_mels+2]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))            # [n_freqs, n_mels]
    enorm = 2.0 / (f_pts[2:] - f_pts[:-2])
    fb = fb * enorm[None, :]
    return np.ascontiguousarray(fb.T.astype(np.float32))


def hann_symmetric(n: int) -> np.ndarray:
    """torch.hann_window(n, periodic=False) (transform.py:175)."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * math.pi * k / (n - 1))).astype(np.float32)


def padded_window(cfg: FrontendConfig) -> np.ndarray:
    """The win_length window centred inside n_fft zeros, as torch.stft does."""
    w = np.zeros(cfg.n_fft, dtype=np.float32)
    left = (cfg.n_fft - cfg.n_window_size) // 2
    w[left:left + cfg.n_window_size] = hann_symmetric(cfg.n_window_size)
    return w


def preemphasis(x: torch.Tensor, coeff: float) -> torch.Tensor:
    """y[0] = x[0]; y[n] = x[n] - coeff * x[n-1].  Not length-masked (transform.py:136-144)."""
    y = x.clone()
    y[:, 1:] = x[:, 1:] - coeff * x[:, :-1]
    return y


def feature_lengths(lengths: torch.Tensor, hop: int) -> torch.Tensor:
    """floor(len / hop) + 1 as int64 (transform.py:182-184)."""
    return (torch.floor(lengths / hop) + 1).to(torch.long)


def power_spectrum(x: torch.Tensor, cfg: FrontendConfig, dtype=torch.float32) -> torch.Tensor:
    """|STFT|^2, shape [B, n_freqs, frames], frames = T // hop + 1 (transform.py:186-208).

    centre=True: reflect-pad n_fft//2 each side; frame f covers padded samples
    [f*hop, f*hop + n_fft)."""
    B, T = x.shape
    half = cfg.n_fft // 2
    xp = torch.nn.functional.pad(x.to(dtype).unsqueeze(1), (half, half), mode="reflect").squeeze(1)
    frames = xp.unfold(-1, cfg.n_fft, cfg.n_window_stride)          # [B, F, n_fft]
    win = torch.from_numpy(padded_window(cfg)).to(dtype)
    spec = torch.fft.rfft(frames * win, dim=-1)                     # [B, F, n_freqs]
    power = spec.real ** 2 + spec.imag ** 2
    return power.transpose(1, 2).contiguous()


def log_mel(power: torch.Tensor, cfg: FrontendConfig) -> torch.Tensor:
    """log(fb @ P + 2^-24) (transform.py:243-255)."""
    fb = torch.from_numpy(slaney_mel_filterbank(cfg.n_freqs, cfg.nfilt, cfg.sample_rate)).to(power.dtype)
    return torch.log(torch.matmul(fb.unsqueeze(0), power) + LOG_FLOOR)


def filterbank_features(x: torch.Tensor, lengths: torch.Tensor, cfg: FrontendConfig = FrontendConfig(),
                        dtype=torch.float32, return_stages: bool = False):
    """Whole eval-mode front end: [B, T] waveform -> ([B, nfilt, frames] features, frame lengths).

    Mirrors FilterbankFeatures (transform.py:288-321) in eval mode (dither off).  `dtype` float64
    gives a higher-precision truth for tolerance studies; the fixtures use float32 like the
    reference."""
    x = x.to(dtype)
    pe = preemphasis(x, cfg.preemph)
    power = power_spectrum(pe, cfg, dtype)
    flen = feature_lengths(lengths, cfg.n_window_stride)
    lm = log_mel(power, cfg)
    mask = lengths_to_mask(flen, lm.shape[-1]).unsqueeze(1)
    feats = masked_normalize(lm, mask, div_guard=1e-5)
    if return_stages:
        return {"preemph": pe, "power": power, "logmel": lm, "features": feats, "lengths": flen}
    return feats, flen
