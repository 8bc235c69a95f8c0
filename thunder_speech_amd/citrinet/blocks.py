"""Citrinet encoder -- module tree, constructor signatures and state-dict keys of the reference's
src/thunder/citrinet/blocks.py (SqueezeExcite :48-83, CitrinetBlock :86-197, stem :200-216, body :219-255,
CitrinetEncoder :258-278).

The sub-block kernel already covers Citrinet's geometry (stride only on the last repeat, residual stride = stride).
The squeeze-excite launch sequence (pool over ALL frames incl. padding -- quirk A3 --, gate MLP, gate * x + residual,
ReLU) is SURVEY section 8 config C3 and is the next row to be built; until then `CitrinetBlock.forward` fails loudly
rather than running anything that is not a HIP kernel.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
from torch import nn
from torch.nn.common_types import _size_1_t

from ..blocks import Masked, MultiSequential, _PackedCache, get_same_padding
from ..quartznet.blocks import EncoderSequential, _FusedBlockBase, _get_act_dropout_layer, _get_conv_bn_layer

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
        raise NotImplementedError("SqueezeExcite: HIP kernels for the Citrinet SE path are not built yet (DESIGN.md §7)")


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

    def forward(self, x: torch.Tensor, lengths: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError("CitrinetBlock: the squeeze-excite HIP launch sequence is not built yet "
                                  "(SURVEY section 8 config C3, DESIGN.md §7); no fallback is provided")


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
