"""asr_collate -- reference src/thunder/data/dataloader_utils.py:17-33 with the padding done by one HIP launch.

The reference sorts the samples by length (longest first), pads them with torch.nn.utils.rnn.pad_sequence (one ATen copy
per clip) and returns float lengths.  Here the ragged clips are gathered into the padded batch by ts_collate_pad: clips that
are still on the host are packed into ONE pinned staging buffer and cross the bus in one copy; clips that already live on
the GPU (e.g. the output of AudioFileLoader.preprocess_audio) are read in place through a pointer table."""
from __future__ import annotations

from typing import List, Tuple

import torch
from torch import Tensor

from .. import _lib

__all__ = ["asr_collate"]


def asr_collate(samples: List[Tuple[Tensor, str]], device="cuda") -> Tuple[Tensor, Tensor, List[str]]:
    """-> (padded audios f32 [B, T_max] on `device`, lengths f32 [B] on `device`, texts), sorted longest first like the
    reference (Python's stable sort, so ties keep their input order)."""
    samples = sorted(samples, key=lambda sample: sample[0].size(-1), reverse=True)
    clips = [s[0].reshape(-1) for s in samples]                       # .squeeze() of [1, T] / [T]
    lens = [int(c.numel()) for c in clips]
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("asr_collate: thunder_speech_amd collates on the GPU only (no CPU fallback)")
    t_max = lens[0]
    keep = []                                                             # device buffers the pointer table refers to
    ptrs = []
    host = [i for i, c in enumerate(clips) if not c.is_cuda]
    if host:
        total = sum(lens[i] for i in host)
        stage = torch.empty(total, dtype=torch.float32, pin_memory=True)
        off, offsets = 0, {}
        for i in host:
            stage[off:off + lens[i]].copy_(clips[i])
            offsets[i] = off
            off += lens[i]
        on_dev = stage.to(dev, non_blocking=True)
        keep.append(on_dev)
    for i, c in enumerate(clips):
        if c.is_cuda:
            c = c.to(device=dev, dtype=torch.float32).contiguous()
            keep.append(c)
            ptrs.append(c.data_ptr())
        else:
            ptrs.append(on_dev.data_ptr() + 4 * offsets[i])
    table = torch.tensor([[p, n] for p, n in zip(ptrs, lens)], dtype=torch.int64).to(dev, non_blocking=True)
    out = torch.empty(len(clips), t_max, dtype=torch.float32, device=dev)
    st = _lib.lib().ts_collate_pad(table.data_ptr(), len(clips), t_max, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(st, "ts_collate_pad")
    for k in keep:                                                       # the launch reads them asynchronously
        k.record_stream(torch.cuda.current_stream(dev))
    audio_lengths = torch.tensor([float(n) for n in lens], dtype=torch.float32).to(dev, non_blocking=True)
    return out, audio_lengths, [s[1] for s in samples]
