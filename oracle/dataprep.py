"""TEST INFRASTRUCTURE (CPU oracle) -- audio preparation and collation in front of the hot path.

* preprocess_audio: reference data/dataset.py:49-77 (mono = mean over channels, minus the mean over time, resample).
* resample: third-party arithmetic, torchaudio.functional.resample of torchaudio 0.12.0 (pinned in the reference's
  poetry.lock; absent from this image), restated from its published form (_get_sinc_resample_kernel /
  _apply_sinc_resample_kernel, "sinc_interpolation", lowpass_filter_width=6, rolloff=0.99).  PARITY UNPINNED for the
  resampler: no reference test holds a resampled vector; cross-checked against scipy.signal.resample_poly on a band-limited
  tone in tests/test_oracle_dataprep.py.
* asr_collate: reference data/dataloader_utils.py:17-33, pinned by tests/golden/r2_misc.npz (generated from the imported
  reference function)."""
from __future__ import annotations

import math
from typing import List, Tuple

import torch
import torch.nn.functional as F


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """-> (kernel f32 [new, 1, kw], width, orig, new) with both rates divided by their gcd."""
    gcd = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // gcd, int(new_freq) // gcd
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = t * base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=t.dtype), t.sin() / t)
    kernels = kernels * window * scale
    return kernels.to(torch.float32), width, orig, new


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    if orig_freq == new_freq:
        return waveform
    kernel, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
    shape = waveform.size()
    wav = waveform.reshape(-1, shape[-1])
    num, length = wav.shape
    wav = F.pad(wav, (width, width + orig))
    res = F.conv1d(wav[:, None], kernel, stride=orig)
    res = res.transpose(1, 2).reshape(num, -1)
    target = int(math.ceil(new * length / orig))
    return res[..., :target].reshape(shape[:-1] + (target,))


def preprocess_audio(audio: torch.Tensor, sample_rate: int, force_mono: bool = True, target_rate: int = 16000) -> torch.Tensor:
    """audio [channels, time] -> [1, time'] (dataset.py:63-77)."""
    if force_mono and audio.shape[0] > 1:
        audio = audio.mean(0, keepdim=True)
    audio = audio - audio.mean(1)
    if target_rate != sample_rate:
        audio = resample(audio, int(sample_rate), int(target_rate))
    return audio


def asr_collate(samples: List[Tuple[torch.Tensor, str]]):
    samples = sorted(samples, key=lambda s: s[0].size(-1), reverse=True)
    t_max = samples[0][0].size(-1)
    out = torch.zeros(len(samples), t_max, dtype=samples[0][0].dtype)
    for i, (a, _) in enumerate(samples):
        out[i, : a.size(-1)] = a.reshape(-1)
    return out, torch.tensor([float(s[0].size(-1)) for s in samples]), [s[1] for s in samples]
