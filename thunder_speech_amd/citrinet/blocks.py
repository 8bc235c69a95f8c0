"""Citrinet encoder -- module tree, constructor signatures and state-dict keys of the reference's
src/thunder/citrinet/blocks.py (SqueezeExcite :48-83, CitrinetBlock :86-197, stem :200-216, body :219-255,
CitrinetEncoder :258-278).

Launch sequence of a block (eval): R fused sub-block launches (the last one without ReLU or residual) -> squeeze-excite
gate (`ts_se_gate_fwd`: pool over ALL frames incl. padding -- quirk A3 --, two bias-free linears, sigmoid) -> the residual
1x1 conv + BN as a pointwise launch -> `ts_se_apply_fwd`: relu(gate * y + res).  All of it runs in HIP; there is no
fallback.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
from torch import nn
from torch.nn.common_types import _size_1_t

from .. import _lib, plan as _plan, tensors as _t
from ..blocks import Masked, MultiSequential, _PackedCache, get_same_padding
from ..quartznet.blocks import EncoderSequential, _FusedBlockBase, _bn_tensors, _get_act_dropout_layer, _get_conv_bn_layer

# the squeeze-excite tail inside the residual launch's epilogue (ABI v7); False = always the separate ts_se_apply_fwd pass (A/B, tests)
FUSE_SE_TAIL = True

__all__ = ["SqueezeExcite", "CitrinetBlock", "stem", "body", "CitrinetEncoder"]


class SqueezeExcite(nn.Module):
    def __init__(self, channels: int, reduction_ratio: int):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool1d(1)
        self.fc = nn.Sequential(
            nn.Linear(channels, channels // reduction_ratio, bias=False),
            nn.ReLU(True),
            nn.Linear(channels // reduction_ratio, channels, bias=False),
        )

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: [batch, channels, time] -> x * sigmoid(fc(mean_t x)); the mean runs over every frame (quirk A3)."""
        _t.require_gpu(x, "SqueezeExcite")
        b, c, t = x.shape
        dev = x.device
        full = torch.full((b,), t, dtype=torch.int32, device=dev)
        xi = _t.backing(x) if _t.is_internal(x) else _t.backing(_t.pack(x, full, slot=("se", id(self))))
        zeros = torch.zeros(c, dtype=torch.float32, device=dev)
        gate = se_gate(self, xi, full, zeros, t)
        out = _t.alloc(b, c, t, dev)
        se_apply(xi, None, gate, full, zeros, None, t, out, relu=False, zero_tail=False)
        y = out[:, :, :t]
        return y if _t.is_internal(x) else _t.unpack(y)


def se_gate(se: SqueezeExcite, y: torch.Tensor, len_i32: torch.Tensor, tail_y: torch.Tensor, t: int) -> torch.Tensor:
    """y: bf16 [B, C, pitch] backing buffer with zeroed tails -> gate f32 [B, C] (ts_se_gate_fwd)."""
    b, c, pitch = y.shape
    w1 = se.fc[0].weight.detach().to(device=y.device, dtype=torch.float32).contiguous()
    w2 = se.fc[2].weight.detach().to(device=y.device, dtype=torch.float32).contiguous()
    pool = torch.empty(b * (c + w1.shape[0]), dtype=torch.float32, device=y.device)      # means + hidden activations
    gate = torch.empty(b, c, dtype=torch.float32, device=y.device)
    st = _lib.lib().ts_se_gate_fwd(y.data_ptr(), len_i32.data_ptr(), tail_y.data_ptr(), b, c, t, pitch, w1.shape[0],
                                   w1.data_ptr(), w2.data_ptr(), pool.data_ptr(), gate.data_ptr(),
                                   torch.cuda.current_stream(y.device).cuda_stream)
    _lib.check(st, "ts_se_gate_fwd")
    return gate


def se_apply(y, r, gate, len_i32, tail_y, tail_r, t: int, out, relu: bool, zero_tail: bool, r_stride: int = 1) -> None:
    """out = act(gate * y + r[..., ::r_stride]) through ts_se_apply_fwd; y / r / out are [B, C, pitch] backing buffers."""
    b, c, _ = y.shape
    st = _lib.lib().ts_se_apply_fwd(y.data_ptr(), None if r is None else r.data_ptr(), gate.data_ptr(), len_i32.data_ptr(),
                                    tail_y.data_ptr(), None if r is None else tail_r.data_ptr(), b, c, t, y.shape[2],
                                    0 if r is None else r.shape[2], int(r_stride), out.shape[2], int(relu), int(zero_tail),
                                    out.data_ptr(), torch.cuda.current_stream(y.device).cuda_stream)
    _lib.check(st, "ts_se_apply_fwd")


class CitrinetBlock(_FusedBlockBase):
    def __init__(self, in_channels: int, out_channels: int, repeat: int = 5, kernel_size: _size_1_t = (11,),
                 stride: _size_1_t = (1,), dilation: _size_1_t = (1,), dropout: float = 0.0, residual: bool = True,
                 separable: bool = False):
        super().__init__()
        padding_val = get_same_padding(kernel_size[0], 1, dilation[0])
        inplanes_loop = in_channels
        conv = []
        for _ in range(repeat - 1):
            conv.extend(_get_conv_bn_layer(inplanes_loop, out_channels, kernel_size=kernel_size, stride=(1,),
                                           dilation=dilation, padding=padding_val, separable=separable, bias=False))
            conv.extend(_get_act_dropout_layer(drop_prob=dropout))
            inplanes_loop = out_channels
        padding_val = get_same_padding(kernel_size[0], stride[0], dilation[0])
        conv.extend(_get_conv_bn_layer(inplanes_loop, out_channels, kernel_size=kernel_size, stride=stride,
                                       dilation=dilation, padding=padding_val, separable=separable, bias=False))
        conv.append(Masked(SqueezeExcite(out_channels, reduction_ratio=8)))
        self.mconv = MultiSequential(*conv)
        if residual:
            stride_residual = stride if stride[0] == 1 else stride[0]
            self.res = MultiSequential(*_get_conv_bn_layer(in_channels, out_channels, kernel_size=1,
                                                           stride=stride_residual, bias=False))
        else:
            self.res = None
        self.mout = MultiSequential(*_get_act_dropout_layer(drop_prob=dropout))
        self.separable = separable
        self.repeat = repeat
        self._cache = _PackedCache()

    def _has_se(self) -> bool:
        return True

    def _params(self) -> List[torch.Tensor]:
        se = self.mconv[len(self.mconv) - 1].layer[0]
        return super()._params() + [se.fc[0].weight, se.fc[2].weight]

    def _compile(self):
        layers = super()._compile()          # the last sub-block comes without ReLU / residual (see _has_se)
        if self.res is not None:
            rc = self.res[0]
            # a strided 1x1 residual conv runs at the INPUT's frame rate on the fast pointwise kernel (a 1x1 conv commutes with
            # subsampling); the SE-apply launch then reads every `stride`-th frame of it
            layers.append(_plan.make_tcs_layer(rc.conv.weight.device, dw_w=None, pw_w=rc.conv.weight,
                                               bn=_bn_tensors(self.res[1].layer[0]), kernel=1, stride=1, dilation=1,
                                               padding=0, relu=False))
        return layers

    def _run_fused(self, x: torch.Tensor, lengths: torch.Tensor, internal: bool = False, slot=None):
        """Same contract as _FusedBlockBase._run_fused; the tail of the block is the SE launch sequence."""
        _t.require_gpu(x, type(self).__name__)
        self._check_eval()
        slot = (id(self), 0) if slot is None else slot
        layers = self._cache.get(self._params(), self._compile)
        convs = layers[:self.repeat]
        res_layer = layers[self.repeat] if self.res is not None else None
        was_internal = _t.is_internal(x)
        xi = x if was_internal else _t.pack(x, lengths, slot=("blk", id(self)))
        x0_tz = _t.is_tail_zero(xi)
        b, _, t = xi.shape
        x0 = _t.backing(xi)
        dev = xi.device
        len_in = _t.lengths_i32(lengths, dev)
        h, th, lh, h_tz = x0, t, len_in, x0_tz
        out_lengths = lengths
        subs = list(self._sub_blocks())
        for r in range(len(convs)):
            layer = convs[r]
            geom = subs[r][0] if subs[r][0] is not None else subs[r][1]
            out = _t.arena(("enc", slot, r % 2), b, layer.c_out, layer.out_size(th), dev)
            h, th = layer.run(h, th, lh, out=out, in_tail_zero=h_tz, zero_tail=True)
            h_tz = True
            if geom.stride != 1 or 2 * geom.padding != geom.dilation * (geom.kernel_size - 1):
                out_lengths = geom.get_seq_len(out_lengths)
                lh = _t.lengths_i32(out_lengths, dev)
        c_out = convs[-1].c_out
        se = self.mconv[len(self.mconv) - 1].layer[0]
        tail_y = convs[-1].bias[:c_out]
        gate = se_gate(se, h, lh, tail_y, th)
        r_buf = tail_r = None
        r_stride = 1
        if res_layer is not None and internal and self.res[0].stride == 1 and x0_tz and FUSE_SE_TAIL:
            # the block's tail relu(gate * main + residual) inside the residual 1x1 launch's epilogue (ts_tcs_desc.se_y): no residual tensor, no
            # separate pass; blocks the library has no such kernel for (strided, caller-visible) take the three-tensor pass below
            out = _t.arena(("enc", slot, "out"), b, c_out, th, dev)
            if h.shape == out.shape:
                y_f, t_res = res_layer.run(x0, t, len_in, out=out, in_tail_zero=True, zero_tail=True, se_y=h, se_gate=gate)
                if y_f is not None:
                    assert t_res == th
                    y = out[:, :, :th]
                    _t.tag_tail_zero(y)
                    return y, out_lengths, was_internal
        if res_layer is not None:
            r_stride = self.res[0].stride
            r_buf = _t.arena(("enc", slot, "res"), b, c_out, t, dev)
            r_buf, t_res = res_layer.run(x0, t, len_in, out=r_buf, in_tail_zero=x0_tz, zero_tail=True)
            assert (t_res - 1) // r_stride + 1 == th
            tail_r = res_layer.bias[:c_out]
        out = _t.arena(("enc", slot, "out"), b, c_out, th, dev) if internal else _t.alloc(b, c_out, th, dev)
        se_apply(h, r_buf, gate, lh, tail_y, tail_r, th, out, relu=True, zero_tail=internal, r_stride=r_stride)
        y = out[:, :, :th]
        if internal:
            _t.tag_tail_zero(y)
        return y, out_lengths, was_internal

    def forward(self, x: torch.Tensor, lengths: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.training:
            return self._forward_train(x, lengths)
        y, out_lengths, was_internal = self._run_fused(x, lengths)
        return (y if was_internal else _t.unpack(y)), out_lengths


def stem(feat_in: int) -> CitrinetBlock:
    return CitrinetBlock(feat_in, 256, repeat=1, kernel_size=(5,), residual=False, separable=True)


def body(filters: List[int], kernel_size: List[int], strides: List[int], dropout: float = 0.0) -> List[CitrinetBlock]:
    layers = []
    f_in = 256
    for f, k, s in zip(filters, kernel_size, strides):
        layers.append(CitrinetBlock(f_in, f, kernel_size=(k,), stride=(s,), separable=True, dropout=dropout))
        f_in = f
    layers.append(CitrinetBlock(f_in, 640, repeat=1, kernel_size=(41,), residual=False, separable=True, dropout=dropout))
    return layers


def CitrinetEncoder(filters: List[int], kernel_sizes: List[int], strides: List[int], feat_in: int = 80,
                    dropout: float = 0.0) -> nn.Module:
    return EncoderSequential(stem(feat_in), *body(filters, kernel_sizes, strides, dropout))
