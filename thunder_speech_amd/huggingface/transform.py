"""wav2vec2 input normalisation -- reference API of src/thunder/huggingface/transform.py:15-55."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import nn

from .. import _lib, tensors as _t

__all__ = ["Wav2Vec2Preprocess"]


class Wav2Vec2Preprocess(nn.Module):
    def __init__(self, div_guard: float = 1e-7, mask_input: bool = False):
        """div_guard: guard against division by zero; mask_input: normalise over the valid samples only and zero the rest
        (models trained with an attention mask); `wav2vec2-large-960h` uses mask_input=False."""
        super().__init__()
        self.div_guard = div_guard
        self.mask_input = mask_input

    def forward(self, audio: torch.Tensor, audio_lengths: torch.Tensor) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """audio [batch, time] -> (normalised audio [batch, time] fp32, audio_lengths) through ts_w2v_preprocess."""
        _t.require_gpu(audio, "Wav2Vec2Preprocess")
        x = audio.to(torch.float32).contiguous()
        b, n = x.shape
        L = _lib.lib()
        li = _t.lengths_i32(audio_lengths, x.device) if self.mask_input else None
        ws = torch.empty(L.ts_w2v_workspace_bytes(b), dtype=torch.uint8, device=x.device)
        out = torch.empty_like(x)
        st = L.ts_w2v_preprocess(x.data_ptr(), li.data_ptr() if li is not None else None, b, n, int(self.mask_input),
                                 float(self.div_guard), out.data_ptr(), ws.data_ptr(),
                                 torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(st, "ts_w2v_preprocess")
        return out, audio_lengths
