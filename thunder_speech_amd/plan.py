"""Host-side compilation of a reference module tree into fused-kernel launch plans.

The reference executes every sub-block as ~7 ATen ops (quartznet/blocks.py:166-182, :222, :281).  Here a
sub-block (depthwise K -> pointwise -> BN -> [+residual] -> ReLU) is ONE launch of
`ts_tcs_subblock_fwd`; this module owns the parameter transformations that launch needs:

  * `fold_bn`        eval-mode BatchNorm1d(eps=1e-3) -> per-channel (scale, shift)      (A12)
  * `pack_dw_taps`   depthwise taps -> pre-shifted Toeplitz rows for v_mfma_f32_4x4x4_16b_bf16
  * `pack_pw_frags`  pointwise weights * scale -> per-lane B fragments of v_mfma_f32_32x32x16_bf16

All packers are pure tensor index arithmetic (run on any device, unit-tested on CPU); the arithmetic on
activations happens only in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib

KC = 64          # input-channel chunk of the kernel
NKP = 3          # depthwise k-steps per pass (dw_ksteps must be a multiple)
BN_EPS = 1e-3


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def fold_bn(weight, bias, running_mean, running_var, eps: float = BN_EPS):
    scale = weight / torch.sqrt(running_var + eps)
    return scale, bias - running_mean * scale


def dw_ksteps(kernel: int, stride: int, dilation: int, padding: int) -> int:
    padl4 = round_up(padding, 4)
    d = padl4 - padding
    vmax = 3 * stride + (kernel - 1) * dilation + d
    return round_up(vmax // 4 + 1, NKP)


def pack_dw_taps(w: torch.Tensor, stride: int, dilation: int, padding: int) -> Tuple[torch.Tensor, int]:
    """w: [C, 1, K] depthwise taps -> bf16 [C_pad64, 4, 4*NK].

    Row i (0..3) of channel c is the Toeplitz row of output frame i of a 4-frame step:
        out[t + i] = sum_v row_i[v] * xwin[v],   xwin[v] = x[t*stride - padl4 + v]
    i.e. row_i[v] = w[u] with v = i*stride + u*dilation + (padl4 - padding), 0 elsewhere."""
    c, _, k = w.shape
    nk = dw_ksteps(k, stride, dilation, padding)
    d = round_up(padding, 4) - padding
    out = torch.zeros(round_up(c, KC), 4, 4 * nk, dtype=torch.float32, device=w.device)
    for i in range(4):
        for u in range(k):
            out[:c, i, i * stride + u * dilation + d] = w[:, 0, u]
    return out.to(torch.bfloat16).contiguous(), nk


def pack_dw_taps_raw(w: torch.Tensor, padding: int) -> torch.Tensor:
    """w: [C, 1, K] depthwise taps of a stride-1, dilation-1 conv -> the RAW tap image of the split kernel, bf16
    [C_pad64/64][4][TAPB/2]: per 16-channel group a KiB-padded block, per channel CST bytes (16 NK + 16 rounded up to 16 mod 32,
    so that a half-wave's reads fall on 32 different LDS banks) holding two copies of the zero-padded array
    wp[n] = w[n - 3 - d] (d = round_up(padding, 4) - padding, n < 4 NK + 4), the second shifted by one element, with their
    DWORDS INTERLEAVED: dword j of copy c sits at byte 8 j + 4 c.  Toeplitz row i of k-step k (pack_dw_taps: row_i[v] =
    w[v - i - d]) is wp[4k + 3 - i .. 4k + 6 - i] = dwords 2k + (i < 2) and the following one of copy (i even): half the bytes
    of the four pre-shifted rows."""
    c, _, k = w.shape
    nk = dw_ksteps(k, 1, 1, padding)
    d = round_up(padding, 4) - padding
    n = 4 * nk + 4
    cp = round_up(c, KC)
    wp = torch.zeros(cp, n + 1, dtype=torch.float32, device=w.device)
    wp[:c, 3 + d: 3 + d + k] = w[:, 0, :]
    pairs = torch.stack([wp[:, :n].reshape(cp, n // 2, 2), wp[:, 1: n + 1].reshape(cp, n // 2, 2)], dim=2)   # [C, dword j, copy, 2]
    cst = raw_tap_channel_stride(nk)
    chan = torch.zeros(cp, cst // 2, dtype=torch.float32, device=w.device)
    chan[:, : 2 * n] = pairs.reshape(cp, 2 * n)
    tapb = (16 * cst + 1023) // 1024 * 1024
    out = torch.zeros(cp // 16, tapb // 2, dtype=torch.float32, device=w.device)
    out[:, : 16 * cst // 2] = chan.reshape(cp // 16, 16 * cst // 2)
    return out.to(torch.bfloat16).reshape(cp // KC, 4, tapb // 2).contiguous()


def raw_tap_channel_stride(nk: int) -> int:
    """Bytes per channel of the raw tap image (mirror of CST in csrc/tcs_kernel.hip)."""
    return 16 * nk + 16 if (16 * nk + 16) % 32 == 16 else 16 * nk + 32


def tap_fragments(taps: torch.Tensor) -> torch.Tensor:
    """[C_pad64, 4, 4*NK] Toeplitz rows -> the order the producer waves load them in:
    [chunk(64 ch)][wave(16 ch)][k-step][lane = 4*(ch % 16) + row][4 samples], so that one wave-instruction reads
    512 contiguous bytes (the row-major order gives 64 different cache lines per instruction)."""
    cp, _, n4 = taps.shape
    nk = n4 // 4
    t = taps.view(cp // KC, 4, 16, 4, nk, 4)            # [chunk][wave][ch][row][k][4]
    return t.permute(0, 1, 4, 2, 3, 5).contiguous().view(cp // KC, 4, nk, 64, 4)


def pack_pw_frags(wf: torch.Tensor) -> torch.Tensor:
    """wf: [Cout, Cin] (already multiplied by the BN scale) -> bf16 [Cout_pad32/32, Cin_pad64/16, 64, 8].

    Fragment (cot, ks), lane l (n = l & 31, h = l >> 5), element j  =  W[cot*32 + n][ks*16 + 8h + j]:
    the B operand of v_mfma_f32_32x32x16_bf16 for D[t][co] += dw[t][ci] * W[co][ci]."""
    cout, cin = wf.shape
    cop, cip = round_up(cout, 32), round_up(cin, KC)
    wp = torch.zeros(cop, cip, dtype=torch.float32, device=wf.device)
    wp[:cout, :cin] = wf
    fr = wp.view(cop // 32, 32, cip // 16, 2, 8).permute(0, 2, 3, 1, 4)      # [cot][ks][h][n][j]
    return fr.reshape(cop // 32, cip // 16, 64, 8).to(torch.bfloat16).contiguous()


def pack_pw_frags16(wf: torch.Tensor) -> torch.Tensor:
    """wf: [Cout, Cin] (already multiplied by the BN scale) -> bf16 [Cout_pad32/16, Cin_pad64/32, 64, 8].

    Fragment (tile, ks), lane l (n = l & 15, g = l >> 4), element j  =  W[tile*16 + n][ks*32 + 8g + j]:
    the B operand of v_mfma_f32_16x16x32_bf16 for D[t][co] += dw[t][ci] * W[co][ci] (the split kernel's consumers)."""
    cout, cin = wf.shape
    cop, cip = round_up(cout, 32), round_up(cin, KC)
    wp = torch.zeros(cop, cip, dtype=torch.float32, device=wf.device)
    wp[:cout, :cin] = wf
    fr = wp.view(cop // 16, 16, cip // 32, 4, 8).permute(0, 2, 3, 1, 4)      # [tile][ks][g][n][j]
    return fr.reshape(cop // 16, cip // 32, 64, 8).to(torch.bfloat16).contiguous()


def pad_bias(b: torch.Tensor) -> torch.Tensor:
    out = torch.zeros(round_up(b.shape[0], 32), dtype=torch.float32, device=b.device)
    out[: b.shape[0]] = b
    return out


def conv_out_size(t: int, kernel: int, stride: int, padding: int, dilation: int) -> int:
    return (t + 2 * padding - dilation * (kernel - 1) - 1) // stride + 1


@dataclass
class TcsLayer:
    """One fused launch: parameters already packed on the target device."""
    c_in: int
    c_out: int
    kernel: int
    stride: int
    dilation: int
    padding: int
    depthwise: bool
    relu: bool
    taps: Optional[torch.Tensor]
    nk: int
    pw: torch.Tensor
    bias: torch.Tensor
    c_res: int = 0
    res_w: Optional[torch.Tensor] = None
    res_stride: int = 1
    pw16: Optional[torch.Tensor] = None            # the pointwise / residual weights as 16x16x32 fragments (split kernel; pack_pw_frags16)
    res_w16: Optional[torch.Tensor] = None
    out_fp32: bool = False
    taps_phase: Optional[torch.Tensor] = None      # dilation 2: the same taps packed for the phase-split kernel
    nk_phase: int = 0
    taps_raw: Optional[torch.Tensor] = None        # stride 1: raw tap image of the split kernel (pack_dw_taps_raw)
    taps_phase_raw: Optional[torch.Tensor] = None
    # Convolutions without a fused kernel of their own (dense K > 1, depthwise stride > 2): `pre` = (K, stride, dilation, padding,
    # source channels, masked) describes an im2col pass (ts_im2col_time) in front of this -- then pointwise-only -- layer, whose c_in is
    # K * source channels.  masked: the reference re-masks between a depthwise and its pointwise conv, so the pointwise input counts as
    # zero beyond the OUTPUT length; a dense conv has no such mask (frames beyond the length keep their partial sums).
    pre: Optional[Tuple[int, int, int, int, int, bool]] = None

    def out_size(self, t_in: int) -> int:
        if self.pre is not None:
            k, s, d, p = self.pre[:4]
            return conv_out_size(t_in, k, s, p, d)
        return conv_out_size(t_in, self.kernel, self.stride, self.padding, self.dilation)

    def desc(self, b: int, t_in: int, pitch_in: int, pitch_out: int, in_tail_zero: bool, zero_tail: bool,
             pitch_res: int = 0, t_res: int = 0) -> "_lib.TcsDesc":
        """struct ts_tcs_desc of this layer for a [b, c_in, pitch_in] input with t_in frames."""
        d = _lib.TcsDesc()
        d.batch, d.c_in, d.c_out, d.t_in, d.t_out = b, self.c_in, self.c_out, t_in, self.out_size(t_in) if self.pre is None else t_in
        d.pitch_in, d.pitch_out = pitch_in, pitch_out
        d.kernel, d.stride, d.dilation, d.padding = self.kernel, self.stride, self.dilation, self.padding
        d.depthwise, d.relu, d.out_fp32 = int(self.depthwise), int(self.relu), int(self.out_fp32)
        d.c_res = self.c_res
        d.res_stride = self.res_stride
        if self.c_res:
            d.pitch_res, d.t_res = pitch_res, t_res
            d.res_w = self.res_w.data_ptr()
            d.res_w16 = self.res_w16.data_ptr() if self.res_w16 is not None else None
        d.dw_ksteps = self.nk
        d.flags = (_lib.TCS_IN_TAILZERO if in_tail_zero else 0) | (_lib.TCS_OUT_ZERO_TAIL if zero_tail else 0)
        d.dw_taps = self.taps.data_ptr() if self.taps is not None else None
        d.dw_taps_raw = self.taps_raw.data_ptr() if self.taps_raw is not None else None
        d.pw_w = self.pw.data_ptr()
        d.pw_w16 = self.pw16.data_ptr() if self.pw16 is not None else None
        d.bias = self.bias.data_ptr()
        return d

    def run(self, x: torch.Tensor, t_in: int, len_in: torch.Tensor, x_res: Optional[torch.Tensor] = None,
            t_res: int = 0, len_res: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
            in_tail_zero: bool = False, zero_tail: bool = False, se_y: Optional[torch.Tensor] = None,
            se_gate: Optional[torch.Tensor] = None):
        """x: bf16 [B, c_in, pitch]; len_in int32 [B].  Returns (y [B, c_out, pitch_out], t_out).
        in_tail_zero: x (and x_res) satisfy the tail-zero invariant (tensors.py) -> mask-free kernels;
        zero_tail: store 0 for frames >= the output length so that y satisfies it too.
        se_y / se_gate: the squeeze-excite tail in this launch's epilogue, y = relu(se_gate * se_y + result) (ts_tcs_desc.se_y); returns
        (None, t_out) -- nothing launched -- when the library has no such kernel for the configuration."""
        if not x.is_cuda:
            raise RuntimeError("thunder_speech_amd kernels run on the GPU only (no CPU fallback)")
        L = _lib.lib()
        b = x.shape[0]
        t_out = self.out_size(t_in)
        if self.pre is not None:
            from . import tensors as _t
            k, s, dl, p, c_src, masked = self.pre
            xcol = _t.arena(("im2col", id(self)), b, k * c_src, t_out, x.device)
            st = L.ts_im2col_time(x.data_ptr(), len_in.data_ptr(), xcol.data_ptr(), b, c_src, t_in, x.shape[2], k, s, dl, p, t_out,
                                  xcol.shape[2], torch.cuda.current_stream(x.device).cuda_stream)
            _lib.check(st, "ts_im2col_time")
            # the pointwise launch masks its input at `len_in`: the output length when the reference masks there (or the caller wants
            # zeroed tails anyway), every frame otherwise
            len_in = _t.lengths_map(len_in, 2 * p - dl * (k - 1) - 1, s, 1) if (masked or zero_tail) else _t.full_lengths(b, t_out, x.device)
            x, t_in, in_tail_zero = xcol, t_out, False
        pitch_out = _lib.time_pitch(t_out)
        if out is None:
            out = torch.empty(b, self.c_out, pitch_out, device=x.device,
                              dtype=torch.float32 if self.out_fp32 else torch.bfloat16)
        d = self.desc(b, t_in, x.shape[2], out.shape[2], in_tail_zero, zero_tail,
                      pitch_res=x_res.shape[2] if self.c_res else 0, t_res=t_res)
        d.t_out = t_out
        stream = torch.cuda.current_stream(x.device).cuda_stream
        args = (x.data_ptr(), len_in.data_ptr(), x_res.data_ptr() if self.c_res else None,
                len_res.data_ptr() if self.c_res else None, out.data_ptr(), stream)
        if se_y is not None:
            if se_gate is None or se_y.shape != out.shape or se_y.dtype != torch.bfloat16 or self.pre is not None:
                return None, t_out
            d.se_y, d.se_gate = se_y.data_ptr(), se_gate.data_ptr()
            st = L.ts_tcs_subblock_fwd(C.byref(d), *args)
            if st == _lib.TS_EUNSUPPORTED:
                return None, t_out
            _lib.check(st, "ts_tcs_subblock_fwd")
            return out, t_out
        if self.taps_phase is not None and in_tail_zero and zero_tail:
            # dilation 2: offer the phase-split fragments first; the library declines geometries it has no such kernel for
            d.flags |= _lib.TCS_TAPS_PHASE
            d.dw_taps, d.dw_ksteps = self.taps_phase.data_ptr(), self.nk_phase
            d.dw_taps_raw = self.taps_phase_raw.data_ptr()
            st = L.ts_tcs_subblock_fwd(C.byref(d), *args)
            if st != _lib.TS_EUNSUPPORTED:
                _lib.check(st, "ts_tcs_subblock_fwd")
                return out, t_out
            d.flags &= ~_lib.TCS_TAPS_PHASE
            d.dw_taps, d.dw_ksteps = self.taps.data_ptr(), self.nk
            d.dw_taps_raw = None
        st = L.ts_tcs_subblock_fwd(C.byref(d), *args)
        _lib.check(st, "ts_tcs_subblock_fwd")
        return out, t_out


def make_im2col_layer(device, *, w2: torch.Tensor, src_channels: int, kernel: int, stride: int, dilation: int, padding: int,
                       masked: bool, **kw) -> TcsLayer:
    """A convolution as im2col + ONE pointwise launch.  w2: [Cout, kernel * src_channels], column u * src_channels + c = the weight of
    tap u of input channel c (dense conv: W[co][c][u]; separable pair: pw[co][c] * dw[c][u]).  `kw`: bn / relu / residual arguments of
    make_tcs_layer."""
    layer = make_tcs_layer(device, dw_w=None, pw_w=w2, kernel=1, stride=1, dilation=1, padding=0, **kw)
    layer.pre = (int(kernel), int(stride), int(dilation), int(padding), int(src_channels), bool(masked))
    return layer


def dense_as_pointwise(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, K] -> [Cout, K * Cin], column u * Cin + c (the row order of ts_im2col_time)."""
    return w.detach().permute(0, 2, 1).reshape(w.shape[0], -1)


def separable_as_pointwise(dw_w: torch.Tensor, pw_w: torch.Tensor) -> torch.Tensor:
    """depthwise [C, 1, K] then pointwise [Cout, C, 1] -> [Cout, K * C], column u * C + c = pw[co][c] * dw[c][u]."""
    dw, pw = dw_w.detach()[:, 0, :], pw_w.detach().reshape(pw_w.shape[0], -1)            # [C, K], [Cout, C]
    return (pw[:, None, :] * dw.t()[None, :, :]).reshape(pw.shape[0], -1)


def make_tcs_layer(device, *, dw_w: Optional[torch.Tensor], pw_w: torch.Tensor, bn: Sequence[torch.Tensor],
                   kernel: int, stride: int, dilation: int, padding: int, relu: bool,
                   res_w: Optional[torch.Tensor] = None, res_bn: Optional[Sequence[torch.Tensor]] = None,
                   res_stride: int = 1, bias_extra: Optional[torch.Tensor] = None, out_fp32: bool = False,
                   pack_on_device: bool = False) -> TcsLayer:
    """Build a fused layer from reference-layout fp32 tensors.

    dw_w: [Cin, 1, K] or None (pointwise only); pw_w: [Cout, Cin, 1] or [Cout, Cin]; bn = (weight, bias,
    running_mean, running_var) or None (scale 1, shift 0; `bias_extra` then carries a conv bias)."""
    # packing is index arithmetic: on the host by default (once per set of weights); `pack_on_device` keeps it on the GPU
    # for weights that change every step (the decoder while fine-tuning) -- no host round trip, no synchronisation
    cpu = (lambda t: t.detach().to(dtype=torch.float32)) if pack_on_device else \
          (lambda t: t.detach().to(device="cpu", dtype=torch.float32))
    pw2 = cpu(pw_w).reshape(pw_w.shape[0], pw_w.shape[1])
    cout, cin = pw2.shape
    if bn is not None:
        scale, shift = fold_bn(*[cpu(t) for t in bn])
    else:
        scale, shift = torch.ones(cout, device=pw2.device), torch.zeros(cout, device=pw2.device)
    if bias_extra is not None:
        shift = shift + cpu(bias_extra) * scale
    wf = pw2 * scale[:, None]
    taps, nk = (None, 0)
    if dw_w is not None:
        taps, nk = pack_dw_taps(cpu(dw_w), stride, dilation, padding)
    taps_phase, nk_phase, taps_raw, taps_phase_raw = None, 0, None, None
    if dw_w is not None and stride == 1 and dilation == 1:
        taps_raw = pack_dw_taps_raw(cpu(dw_w), padding).to(device)
    if dw_w is not None and stride == 1 and dilation == 2 and padding % 2 == 0 and res_w is None:
        taps_phase, nk_phase = pack_dw_taps(cpu(dw_w), 1, 1, padding // 2)
        taps_phase = tap_fragments(taps_phase).to(device)
        taps_phase_raw = pack_dw_taps_raw(cpu(dw_w), padding // 2).to(device)
    c_res, res_p, res_p16 = 0, None, None
    if res_w is not None:
        r2 = cpu(res_w).reshape(res_w.shape[0], res_w.shape[1])
        rs, rsh = fold_bn(*[cpu(t) for t in res_bn])
        res_p = pack_pw_frags(r2 * rs[:, None]).to(device)
        res_p16 = pack_pw_frags16(r2 * rs[:, None]).to(device)
        shift = shift + rsh
        c_res = r2.shape[1]
    return TcsLayer(c_in=cin, c_out=cout, kernel=kernel, stride=stride, dilation=dilation, padding=padding,
                    depthwise=dw_w is not None, relu=relu, taps=None if taps is None else tap_fragments(taps).to(device), nk=nk,
                    pw=pack_pw_frags(wf).to(device), bias=pad_bias(shift).to(device), c_res=c_res, res_w=res_p,
                    res_stride=res_stride, out_fp32=out_fp32, taps_phase=taps_phase, nk_phase=nk_phase, taps_raw=taps_raw,
                    taps_phase_raw=taps_phase_raw,
                    pw16=pack_pw_frags16(wf).to(device), res_w16=res_p16)
