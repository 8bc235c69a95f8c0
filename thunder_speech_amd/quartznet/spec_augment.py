"""SpecAugment / SpecCutout -- reference names and constructor signatures of src/thunder/quartznet/spec_augment.py:23-102.

The reference zeroes segments of the normalised features with one masked_fill pass over [B, F, T] per mask, the same mask
for every clip of the batch; the geometry comes from torch.rand(1) on the host.  Here a mask is a row (f0, f1, t0, t1) of a
small int32 device table:
  * inside FilterbankFeatures the table rides along with the fused front end (csrc/frontend.hip applies it while it writes
    the normalised features: no extra pass over the tensor);
  * a standalone call (`fb[4](x, lengths)` style) zeroes the rectangles in place with ts_spec_mask_apply.
`rng`: "torch" (default) draws the geometry exactly like the reference -- two torch.rand(1) per span from the CPU generator, so
the masks are bit-identical to the reference's under the same torch.manual_seed; "philox" draws them on the device from a
counter-based stream (ts_spec_masks_draw, nothing crosses the bus, graph-capturable)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import _lib
from .. import rng as _rng
from .. import tensors as _t


def _span(mask_param: int, size: int):
    """torchaudio.functional.mask_along_axis (0.12) == spec_augment._create_mask (spec_augment.py:60-75)."""
    value = torch.rand(1) * mask_param
    min_value = torch.rand(1) * (size - value)
    return int(min_value.long()), int(min_value.long() + value.long())


class _MaskTable(nn.Module):
    rng: str = "torch"

    def _counts(self):
        """-> (n_time, time_width, n_freq, freq_width, n_cutout, cut_time_width, cut_freq_width)"""
        raise NotImplementedError

    def n_masks(self) -> int:
        c = self._counts()
        return c[0] + c[2] + c[4]

    def draw(self, n_mels: int, n_frames: int, device) -> Optional[torch.Tensor]:
        """int32 device table [n_masks, 4] = (f0, f1, t0, t1) for one training step, or None when there is nothing to mask."""
        n_time, tw, n_freq, fw, n_cut, ctw, cfw = self._counts()
        n = n_time + n_freq + n_cut
        if n == 0:
            return None
        if self.rng == "philox":
            table = torch.empty(n, 4, dtype=torch.int32, device=device)
            st = _lib.lib().ts_spec_masks_draw(_rng.next_seed(), n_time, tw, n_freq, fw, n_cut, ctw, cfw, n_mels, n_frames,
                                               table.data_ptr(), torch.cuda.current_stream(device).cuda_stream)
            _lib.check(st, "ts_spec_masks_draw")
            return table
        if self.rng != "torch":
            raise ValueError(f"rng must be 'torch' or 'philox', got {self.rng!r}")
        rows = []
        for _ in range(n_cut):                 # spec_augment.py:98-101 (the time span is drawn with freq_width: reference quirk)
            f0, f1 = _span(cfw, n_mels)
            t0, t1 = _span(cfw, n_frames)
            rows.append((f0, f1, t0, t1))
        for _ in range(n_time):                # spec_augment.py:51-53
            t0, t1 = _span(tw, n_frames)
            rows.append((0, n_mels, t0, t1))
        for _ in range(n_freq):                # spec_augment.py:55-56
            f0, f1 = _span(fw, n_mels)
            rows.append((f0, f1, 0, n_frames))
        return torch.tensor(rows, dtype=torch.int32).to(device, non_blocking=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, F, T] (fp32 reference layout or the internal bf16 layout); identity in eval mode."""
        if not self.training or self.n_masks() == 0:
            return x
        _t.require_gpu(x, type(self).__name__)
        # like masked_fill, the result is a new tensor unless x is an internal (library-layout) view, which is masked in place
        xi = x if _t.is_internal(x) else x.to(torch.float32).clone(memory_format=torch.contiguous_format)
        b, f, t = xi.shape
        table = self.draw(f, t, xi.device)
        st = _lib.lib().ts_spec_mask_apply(xi.data_ptr(), xi.element_size(), b, f, t, xi.stride(1), table.data_ptr(), table.shape[0],
                                           torch.cuda.current_stream(xi.device).cuda_stream)
        _lib.check(st, "ts_spec_mask_apply")
        return xi


class SpecAugment(_MaskTable):
    def __init__(self, freq_masks=0, time_masks=0, freq_width=10, time_width=10):
        super().__init__()
        self.freq_masks, self.time_masks = freq_masks, time_masks
        self.freq_width, self.time_width = freq_width, time_width

    def _counts(self):
        return self.time_masks, self.time_width, self.freq_masks, self.freq_width, 0, 0, 0


class SpecCutout(_MaskTable):
    def __init__(self, rect_masks: int = 0, time_width: int = 5, freq_width: int = 20):
        super().__init__()
        self.rect_masks, self.time_width, self.freq_width = rect_masks, time_width, freq_width

    def _counts(self):
        return 0, 0, 0, 0, self.rect_masks, self.time_width, self.freq_width
