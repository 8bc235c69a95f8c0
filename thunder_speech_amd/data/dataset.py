"""AudioFileLoader -- reference src/thunder/data/dataset.py:24-90, with preprocess_audio on the device.

preprocess_audio (dataset.py:49-77): mono mix (mean over channels), DC removal (minus the mean over time), resampling to
the dataset rate.  The reference runs three ATen ops and torchaudio.functional.resample per clip on the host; here the clip
is uploaded once and ts_audio_prep does all three (csrc/dataprep.hip).  The polyphase sinc kernel is torchaudio's
(0.12.0 defaults: sinc_interpolation, lowpass_filter_width 6, rolloff 0.99), built on the host in float64 once per rate pair."""
from __future__ import annotations

import math
import wave
from typing import Dict, Tuple

import numpy as np
import torch
from torch import Tensor, nn

from .. import _lib

__all__ = ["AudioFileLoader"]


def _sinc_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """torchaudio.functional.resample's filter bank: kernel[p][j], p < new, j < 2 * width + orig (rates divided by their gcd)."""
    gcd = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // gcd, new_freq // gcd
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base_freq
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        kern = np.where(t == 0, 1.0, np.sin(t) / t)
    kern = kern * window * (base_freq / orig)
    return torch.from_numpy(kern.astype(np.float32)).contiguous(), width, orig, new


class AudioFileLoader(nn.Module):
    def __init__(self, force_mono: bool = True, sample_rate: int = 16000, device="cuda"):
        super().__init__()
        self.force_mono = force_mono
        self.sample_rate = sample_rate
        self.device = torch.device(device)
        self._kernels: Dict[Tuple[int, int], tuple] = {}

    def open_audio(self, item: str) -> Tuple[Tensor, int]:
        """(channels, time) float audio + sample rate.  torchaudio.load when torchaudio is installed (the reference's reader,
        dataset.py:47), else 16-bit PCM WAV through the standard library."""
        try:
            import torchaudio
            return torchaudio.load(item)
        except ImportError:
            with wave.open(item, "rb") as w:
                if w.getsampwidth() != 2:
                    raise RuntimeError("without torchaudio only 16-bit PCM WAV files can be opened")
                pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, w.getnchannels())
                return torch.from_numpy(pcm.T.astype(np.float32) / 32768.0).contiguous(), w.getframerate()

    def preprocess_audio(self, audio: Tensor, sample_rate: int) -> Tensor:
        """[channels, time] (host or device) -> [1, time'] on the device."""
        if audio.dim() != 2:
            raise ValueError("preprocess_audio expects audio of shape (channels, time)")
        c, t = audio.shape
        if not (self.force_mono or c == 1):
            # the reference's `audio - audio.mean(1)` only broadcasts for one channel (dataset.py:72)
            raise ValueError("multi-channel audio needs force_mono=True")
        x = audio.to(device=self.device, dtype=torch.float32).contiguous()
        sample_rate = int(sample_rate)
        L = _lib.lib()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ws = torch.empty(L.ts_audio_prep_workspace_bytes(t), dtype=torch.uint8, device=self.device)
        if sample_rate == int(self.sample_rate):
            out = torch.empty(1, t, dtype=torch.float32, device=self.device)
            st = L.ts_audio_prep(x.data_ptr(), c, t, None, 1, 1, 0, 0, out.data_ptr(), t, ws.data_ptr(), stream)
        else:
            key = (sample_rate, int(self.sample_rate))
            if key not in self._kernels:
                k, width, orig, new = _sinc_kernel(*key)
                self._kernels[key] = (k.to(self.device), width, orig, new)
            k, width, orig, new = self._kernels[key]
            t_out = int(math.ceil(new * t / orig))
            out = torch.empty(1, t_out, dtype=torch.float32, device=self.device)
            st = L.ts_audio_prep(x.data_ptr(), c, t, k.data_ptr(), orig, new, k.shape[1], width, out.data_ptr(), t_out,
                                 ws.data_ptr(), stream)
        _lib.check(st, "ts_audio_prep")
        return out

    def forward(self, item: str) -> Tensor:
        audio, sample_rate = self.open_audio(item)
        return self.preprocess_audio(audio, sample_rate)
