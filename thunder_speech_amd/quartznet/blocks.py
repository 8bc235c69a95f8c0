"""QuartzNet encoder -- same module tree, constructor signatures and state-dict keys as the reference's
src/thunder/quartznet/blocks.py, executed as fused HIP launches.

Reference tree per block (quartznet/blocks.py:231-315): `mconv` = [dw MaskedConv1d, pw MaskedConv1d, Masked(BN),
Masked(ReLU), Masked(Dropout)] x repeat (the last repeat stops after BN), `res` = [1x1 MaskedConv1d, Masked(BN)],
`mout` = [Masked(ReLU), Masked(Dropout)].  Here each repeat is ONE launch of ts_tcs_subblock_fwd (mask ->
depthwise -> mask -> pointwise -> folded BN -> ReLU), the residual branch and the block's final ReLU are
fused into the last repeat's launch.
"""
from __future__ import annotations

from enum import Enum
from typing import List, Optional, Tuple

import torch
from torch import nn
from torch.nn.common_types import _size_1_t

from .. import plan as _plan
from .. import tensors as _t
from ..blocks import Masked, MultiSequential, _PackedCache, get_same_padding

__all__ = ["InitMode", "init_weights", "MaskedConv1d", "QuartznetBlock", "stem", "body", "QuartznetEncoder"]


class InitMode(str, Enum):
    xavier_uniform = "xavier_uniform"
    xavier_normal = "xavier_normal"
    kaiming_uniform = "kaiming_uniform"
    kaiming_normal = "kaiming_normal"


def init_weights(m: nn.Module, mode: InitMode = InitMode.xavier_uniform):
    """reference quartznet/blocks.py:59-90"""
    if isinstance(m, MaskedConv1d):
        init_weights(m.conv, mode)
    if isinstance(m, (nn.Conv1d, nn.Linear)):
        if mode == InitMode.xavier_uniform:
            nn.init.xavier_uniform_(m.weight, gain=1.0)
        elif mode == InitMode.xavier_normal:
            nn.init.xavier_normal_(m.weight, gain=1.0)
        elif mode == InitMode.kaiming_uniform:
            nn.init.kaiming_uniform_(m.weight, nonlinearity="relu")
        elif mode == InitMode.kaiming_normal:
            nn.init.kaiming_normal_(m.weight, nonlinearity="relu")
        else:
            raise ValueError(f"Unknown Initialization mode: {mode}")
    elif isinstance(m, nn.BatchNorm1d):
        if m.track_running_stats:
            m.running_mean.zero_()
            m.running_var.fill_(1)
            m.num_batches_tracked.zero_()
        if m.affine:
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)


def _conv_len(lengths: torch.Tensor, k: int, s: int, p: int, d: int) -> torch.Tensor:
    return _t.lengths_map(lengths, 2 * p - d * (k - 1) - 1, s, 1)


class MaskedConv1d(nn.Module):
    """Parameter holder with the reference's attributes (quartznet/blocks.py:93-182).  Standalone forward is
    supported for depthwise (groups == channels) and 1x1 convolutions through the fused kernel."""
    __constants__ = ["use_mask", "padding", "dilation", "kernel_size", "stride"]

    def __init__(self, in_channels: int, out_channels: int, kernel_size: _size_1_t, stride: _size_1_t = 1,
                 padding: _size_1_t = 0, dilation: _size_1_t = 1, groups: int = 1, bias: bool = False,
                 use_mask: bool = True):
        super().__init__()
        self.use_mask = use_mask
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                              dilation=dilation, groups=groups, bias=bias)
        self.padding = self.conv.padding[0]
        self.dilation = self.conv.dilation[0]
        self.kernel_size = self.conv.kernel_size[0]
        self.stride = self.conv.stride[0]
        self._cache = _PackedCache()

    def get_seq_len(self, lengths: torch.Tensor) -> torch.Tensor:
        return _conv_len(lengths, self.kernel_size, self.stride, self.padding, self.dilation)

    def mask_fill(self, x: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
        """x with the frames >= lengths[b] zeroed (reference quartznet/blocks.py:158-167).  A tensor helper like lengths_to_mask: plain torch on
        whatever device x lives on; `forward` does not call it -- the kernels apply the mask while they load (or rely on the tail-zero invariant)."""
        from ..blocks import lengths_to_mask
        return x.masked_fill(~lengths_to_mask(lengths, x.shape[-1]).unsqueeze(1), 0)

    def forward(self, x: torch.Tensor, lengths: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        _t.require_gpu(x, "MaskedConv1d")
        conv = self.conv
        depthwise = conv.groups == conv.in_channels == conv.out_channels and conv.groups > 1
        if not depthwise and conv.groups != 1:
            raise NotImplementedError("MaskedConv1d: grouped convolutions other than depthwise have no HIP kernel "
                                      "(the reference builds dense and depthwise ones only, quartznet/blocks.py:185-224)")
        tensors = [conv.weight] + ([conv.bias] if conv.bias is not None else [])

        def build():
            c = conv.out_channels
            bias = None if conv.bias is None else conv.bias.detach()
            if depthwise and self.stride > 2:
                # no fused kernel for this stride: im2col + one pointwise launch over K * C "channels" (weights: identity x taps)
                return _plan.make_im2col_layer(conv.weight.device, w2=_plan.separable_as_pointwise(conv.weight, torch.eye(c, device=conv.weight.device)),
                                               src_channels=c, kernel=self.kernel_size, stride=self.stride, dilation=self.dilation,
                                               padding=self.padding, masked=False, bn=None, relu=False, bias_extra=bias)
            if not depthwise and self.kernel_size > 1:
                # dense K > 1 (quartznet/blocks.py:213-221): im2col + one pointwise launch over K * Cin input channels
                return _plan.make_im2col_layer(conv.weight.device, w2=_plan.dense_as_pointwise(conv.weight), src_channels=conv.in_channels,
                                               kernel=self.kernel_size, stride=self.stride, dilation=self.dilation, padding=self.padding,
                                               masked=False, bn=None, relu=False, bias_extra=bias)
            if depthwise:
                return _plan.make_tcs_layer(conv.weight.device, dw_w=conv.weight.detach(), pw_w=torch.eye(c), bn=None,
                                            kernel=self.kernel_size, stride=self.stride, dilation=self.dilation,
                                            padding=self.padding, relu=False,
                                            bias_extra=None if conv.bias is None else conv.bias.detach())
            return _plan.make_tcs_layer(conv.weight.device, dw_w=None, pw_w=conv.weight.detach(), bn=None, kernel=1,
                                        stride=self.stride, dilation=1, padding=0, relu=False,
                                        bias_extra=None if conv.bias is None else conv.bias.detach())
        layer = self._cache.get(tensors, build)
        if self.use_mask:
            # mask_fill (quartznet/blocks.py:158-167) happens ONCE, on the input: it is packed with the frames >= length zeroed
            # and the launch then runs unmasked, so the frames beyond the output length keep the partial sums the reference's
            # conv leaves there (the fused blocks re-mask between their own convs instead)
            xi = _t.pack(_t.unpack(x), lengths, slot=("mconv", id(self)))      # (the standalone module is not on the hot path)
        else:
            xi = _t.pack(x)
        b, _, t = xi.shape
        full = _t.full_lengths(b, t, xi.device)
        y, t_out = layer.run(_t.backing(xi), t, full)
        return y[:, :, :t_out], self.get_seq_len(lengths)


def _get_conv_bn_layer(in_channels: int, out_channels: int, kernel_size: _size_1_t = 11, separable: bool = False,
                       **conv_kwargs):
    """reference quartznet/blocks.py:185-224"""
    if separable:
        layers = [
            MaskedConv1d(in_channels, in_channels, kernel_size, groups=in_channels, **conv_kwargs),
            MaskedConv1d(in_channels, out_channels, kernel_size=1, stride=1, dilation=1, padding=0,
                         bias=conv_kwargs.get("bias", False)),
        ]
    else:
        layers = [MaskedConv1d(in_channels, out_channels, kernel_size, **conv_kwargs)]
    layers.append(Masked(nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.1)))
    return layers


def _get_act_dropout_layer(drop_prob: float = 0.2):
    return [Masked(nn.ReLU(True)), Masked(nn.Dropout(p=drop_prob))]


def _bn_tensors(bn: nn.BatchNorm1d):
    return [bn.weight, bn.bias, bn.running_mean, bn.running_var]


class _FusedBlockBase(nn.Module):
    """Shared executor of QuartznetBlock / CitrinetBlock: compiles `mconv`/`res` into TcsLayers."""

    separable: bool
    repeat: int

    def _sub_blocks(self):
        """Yield (dw_conv | None, pw_or_dense_conv, bn) for every repeat."""
        step = 5 if self.separable else 4
        for r in range(self.repeat):
            base = r * step
            if self.separable:
                yield self.mconv[base], self.mconv[base + 1], self.mconv[base + 2].layer[0]
            else:
                yield None, self.mconv[base], self.mconv[base + 1].layer[0]

    def _params(self) -> List[torch.Tensor]:
        out = []
        for dw, pw, bn in self._sub_blocks():
            if dw is not None:
                out.append(dw.conv.weight)
            out.append(pw.conv.weight)
            out.extend(_bn_tensors(bn))
        if self.res is not None:
            out.append(self.res[0].conv.weight)
            out.extend(_bn_tensors(self.res[1].layer[0]))
        return out

    def _compile(self) -> List[_plan.TcsLayer]:
        layers = []
        subs = list(self._sub_blocks())
        device = subs[0][1].conv.weight.device
        for r, (dw, pw, bn) in enumerate(subs):
            last = r == len(subs) - 1
            geom = dw if dw is not None else pw
            kw = dict(bn=_bn_tensors(bn), relu=True)
            if last and self._has_se():
                kw["relu"] = False          # SE gate + residual + ReLU follow in separate launches
            elif last and self.res is not None:
                rc = self.res[0]
                kw.update(res_w=rc.conv.weight, res_bn=_bn_tensors(self.res[1].layer[0]), res_stride=rc.stride)
            geo = dict(kernel=geom.kernel_size, stride=geom.stride, dilation=geom.dilation, padding=geom.padding)
            if dw is None and geom.kernel_size != 1:
                # non-separable block with K > 1 (reference-valid, quartznet/blocks.py:213-221; no reference model uses it):
                # im2col + one pointwise launch over K * Cin input channels
                layers.append(_plan.make_im2col_layer(device, w2=_plan.dense_as_pointwise(pw.conv.weight), src_channels=pw.conv.in_channels,
                                                      masked=False, **geo, **kw))
            elif dw is not None and dw.stride > 2:
                # depthwise stride 3+ has no fused kernel: the separable pair as ONE product over K * C input channels
                layers.append(_plan.make_im2col_layer(device, w2=_plan.separable_as_pointwise(dw.conv.weight, pw.conv.weight),
                                                      src_channels=dw.conv.in_channels, masked=True, **geo, **kw))
            else:
                layers.append(_plan.make_tcs_layer(device, dw_w=None if dw is None else dw.conv.weight, pw_w=pw.conv.weight, **geo, **kw))
        return layers

    def _has_se(self) -> bool:
        return False

    def _check_eval(self):
        if self.training:
            raise NotImplementedError(f"{type(self).__name__}: this path is the inference launch sequence; call .eval()")

    def _forward_train(self, x: torch.Tensor, lengths: torch.Tensor):
        """Training-mode forward (batch-statistics BatchNorm, autograd through every op): the reference's op sequence
        (quartznet/blocks.py:317-338), one HIP launch per op and direction (train_ops.py); activations f32 (default) or bf16
        (mixed precision), see train_ops.set_activation_dtype.  Returns activation rows: a [B, C, T] view with a padded pitch."""
        from .. import train_ops as T
        _t.require_gpu(x, type(self).__name__)
        # The training kernels normalise with batch statistics and update the running ones (nn.BatchNorm1d in train mode).  A
        # BatchNorm switched to eval() inside a training-mode block would need running-statistics normalisation with no update:
        # refuse it rather than silently use batch statistics (the reference's fine-tuning schedule never produces this state:
        # BaseFinetuning.freeze flips requires_grad only, callbacks.py)
        if any(isinstance(m, nn.BatchNorm1d) and not m.training for m in self.modules()):
            raise NotImplementedError(f"{type(self).__name__}: a BatchNorm1d in eval mode inside a training-mode block has no HIP "
                                      "kernel; put the whole block in eval() (inference launches) or the BatchNorm in train()")
        # Dropout modules of the reference tree: after the ReLU of every repeat but the last (mconv), and after the block's
        # final ReLU (mout) -- quartznet/blocks.py:227-228; each follows its own training flag like nn.Dropout does
        drops = [m.layer[0] for m in self.mconv if isinstance(m, Masked) and isinstance(m.layer[0], nn.Dropout)]
        out_drop = self.mout[1].layer[0]
        x = T.to_act(x)              # activation rows of the training path's element type (train_ops.set_activation_dtype)
        dev = x.device
        len_in = _t.lengths_i32(lengths, dev)
        if self.res is not None:     # two consumers: one node adds the two gradients on the way back (on activation rows)
            x, x_res = T.Fork.apply(x, len_in if self.res[0].stride == 1 else None)
        h, lh, out_lengths = x, len_in, lengths
        subs = list(self._sub_blocks())
        # block tail relu(BN(main) + BN(residual)) in one pass: both branches then end un-normalised (train_ops.block_tail)
        fuse_tail = (T._LAZY_BN and self.res is not None and self.res[0].stride == 1 and not self._has_se()
                     and not (out_drop.training and out_drop.p > 0.0))
        for r, (dw, pw, bn) in enumerate(subs):
            last = r == len(subs) - 1
            if pw.kernel_size != 1 or pw.stride != 1:
                raise NotImplementedError("training mode: dense convs other than 1x1 / stride 1 have no HIP kernel")
            lh_in = lh
            if dw is not None and (dw.stride != 1 or 2 * dw.padding != dw.dilation * (dw.kernel_size - 1)):
                out_lengths = dw.get_seq_len(out_lengths)          # only length-changing convs cost host work / tiny launches
                lh = _t.lengths_i32(out_lengths, dev)
            # one autograd node per repeat: [depthwise (its output already masked for the pointwise conv) | mask] -> 1x1 -> BN -> ReLU -> Dropout
            drop = drops[r] if (not last and r < len(drops)) else None
            drop_p = drop.p if (drop is not None and drop.training) else 0.0
            # between two repeats the BatchNorm (+ ReLU) is folded into the next repeat's depthwise launches (train_ops.SubBlock)
            lazy = (fuse_tail and drop_p == 0.0) if last else (T._LAZY_BN and drop_p == 0.0 and T.same_depthwise(subs[r + 1][0]))
            h = T.sub_block(h, dw, pw, bn, lh_in, lh, relu=not last, drop_p=drop_p, lazy_out=lazy, tile_stats=not last, defer_stats=last)
        if self._has_se():
            se = self.mconv[len(self.mconv) - 1].layer[0]          # citrinet/blocks.py:154: SE closes the main branch
            h = T.SqueezeExciteTrain.apply(h, se.fc[0].weight, se.fc[2].weight)
        r_out = None
        if self.res is not None:
            rc, rbn = self.res[0], self.res[1].layer[0]
            if rc.stride != 1:       # strided 1x1 MaskedConv1d: mask + subsample, then the pointwise GEMM
                r_in = T.SubsampleMask.apply(x_res, len_in, rc.stride, (x_res.shape[2] - 1) // rc.stride + 1)
                r_out = T.batch_norm_train(rbn, T.PointwiseConv.apply(r_in, rc.conv.weight), relu=False)
            else:                    # mask -> 1x1 -> BatchNorm as one node, like a repeat without depthwise conv and ReLU
                r_out = T.sub_block(x_res, None, rc, rbn, len_in, len_in, relu=False, lazy_out=fuse_tail, bwd_mask=False, defer_stats=True)
        out = T.block_tail(h, r_out) if (fuse_tail and getattr(h, "_ts_lazy", None) is not None) else T.AddRelu.apply(h, r_out)
        return T.dropout(out, out_drop.p, out_drop.training), out_lengths

    def _run_fused(self, x: torch.Tensor, lengths: torch.Tensor, internal: bool = False, slot=None):
        """Run the block's launches.  `internal=True` (set by the encoder for every block but the last): the block
        output is never shown to the caller, so it is written into a library-owned arena buffer with frames >=
        length zeroed (tail-zero invariant) and the next block runs mask-free.  The last sub-block of a
        caller-visible block keeps the reference's values beyond the length (quirk A2) in a fresh buffer."""
        _t.require_gpu(x, type(self).__name__)
        self._check_eval()
        slot = (id(self), 0) if slot is None else slot          # arena buffers are owned by the calling encoder, or by this block
        layers = self._cache.get(self._params(), self._compile)
        was_internal = _t.is_internal(x)
        xi = x if was_internal else _t.pack(x, lengths, slot=("blk", id(self)))
        x0_tz = h_tz = _t.is_tail_zero(xi)
        b, _, t = xi.shape
        x0 = _t.backing(xi)
        len_in = _t.lengths_i32(lengths, xi.device)
        h, th, lh = x0, t, len_in
        out_lengths = lengths
        subs = list(self._sub_blocks())
        n = len(layers)
        for r in range(n):
            layer = layers[r]
            geom = subs[r][0] if subs[r][0] is not None else subs[r][1]
            last = r == n - 1
            keep_tail = last and not internal          # caller-visible: reference values beyond the length
            t_out = layer.out_size(th)
            if keep_tail:
                out = _t.alloc(b, layer.c_out, t_out, xi.device)
            else:
                out = _t.arena(("enc", slot, r % 2), b, layer.c_out, t_out, xi.device)
            kw = dict(out=out, in_tail_zero=h_tz and (x0_tz or not layer.c_res), zero_tail=not keep_tail)
            if layer.c_res:
                kw.update(x_res=x0, t_res=t, len_res=len_in)
            h, th = layer.run(h, th, lh, **kw)
            h_tz = not keep_tail                         # arena buffer + zeroed tail -> invariant holds for the next launch
            if geom.stride != 1 or 2 * geom.padding != geom.dilation * (geom.kernel_size - 1):
                out_lengths = geom.get_seq_len(out_lengths)          # only length-changing convs cost host work
                lh = _t.lengths_i32(out_lengths, xi.device)
        y = h[:, :, :th]
        if internal:
            _t.tag_tail_zero(y)
        return y, out_lengths, was_internal


class QuartznetBlock(_FusedBlockBase):
    def __init__(self, in_channels: int, out_channels: int, repeat: int = 5, kernel_size: _size_1_t = (11,),
                 stride: _size_1_t = (1,), dilation: _size_1_t = (1,), dropout: float = 0.0, residual: bool = True,
                 separable: bool = False):
        """Same arguments as the reference QuartznetBlock (quartznet/blocks.py:231-315)."""
        super().__init__()
        padding_val = get_same_padding(kernel_size[0], stride[0], dilation[0])
        inplanes_loop = in_channels
        conv = []
        for _ in range(repeat - 1):
            conv.extend(_get_conv_bn_layer(inplanes_loop, out_channels, kernel_size=kernel_size, stride=stride,
                                           dilation=dilation, padding=padding_val, separable=separable, bias=False))
            conv.extend(_get_act_dropout_layer(drop_prob=dropout))
            inplanes_loop = out_channels
        conv.extend(_get_conv_bn_layer(inplanes_loop, out_channels, kernel_size=kernel_size, stride=stride,
                                       dilation=dilation, padding=padding_val, separable=separable, bias=False))
        self.mconv = MultiSequential(*conv)
        if residual:
            stride_residual = stride if stride[0] == 1 else stride[0] ** repeat      # A8
            self.res = MultiSequential(*_get_conv_bn_layer(in_channels, out_channels, kernel_size=1,
                                                           stride=stride_residual, bias=False))
        else:
            self.res = None
        self.mout = MultiSequential(*_get_act_dropout_layer(drop_prob=dropout))
        self.separable = separable
        self.repeat = repeat
        self._cache = _PackedCache()

    def forward(self, x: torch.Tensor, lengths: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.training:
            return self._forward_train(x, lengths)
        y, out_lengths, was_internal = self._run_fused(x, lengths)
        return (y if was_internal else _t.unpack(y)), out_lengths


class EncoderSequential(MultiSequential):
    """MultiSequential of fused blocks.  Packs a reference-layout input once (zeroing frames >= length), runs every
    block but the last as `internal` (arena buffers, tail-zero outputs, mask-free kernels) and lets the last block
    produce the caller-visible output with the reference's values beyond the length (quirk A2)."""

    def forward(self, audio: torch.Tensor, audio_lengths: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        _t.require_gpu(audio, "encoder")
        if self.training:                       # training mode: plain chain of the blocks' training forwards (autograd)
            x = audio
            with _t.lengths_scope():
                for blk in self.children():
                    x, audio_lengths = blk(x, audio_lengths)
            return x, audio_lengths
        with _t.lengths_scope():                # the int32 copy of a lengths tensor is made once per forward, not once per block
            x = audio if _t.is_internal(audio) else _t.pack(audio, audio_lengths, slot=("enc", id(self)))
            blocks = list(self.children())
            for i, blk in enumerate(blocks):
                if isinstance(blk, _FusedBlockBase):
                    x, audio_lengths, _ = blk._run_fused(x, audio_lengths, internal=i < len(blocks) - 1, slot=(id(self), i % 2))
                else:
                    x, audio_lengths = blk(x, audio_lengths)
        return x, audio_lengths


def stem(feat_in: int) -> QuartznetBlock:
    return QuartznetBlock(feat_in, 256, repeat=1, stride=(2,), kernel_size=(33,), residual=False, separable=True)


def body(filters: List[int], kernel_size: List[int], repeat_blocks: int = 1, dropout: float = 0.0) -> List[QuartznetBlock]:
    layers = []
    f_in = 256
    for f, k in zip(filters, kernel_size):
        for _ in range(repeat_blocks):
            layers.append(QuartznetBlock(f_in, f, kernel_size=(k,), separable=True, dropout=dropout))
            f_in = f
    layers.extend([
        QuartznetBlock(f_in, 512, repeat=1, dilation=(2,), kernel_size=(87,), residual=False, separable=True,
                       dropout=dropout),
        QuartznetBlock(512, 1024, repeat=1, kernel_size=(1,), residual=False, separable=False, dropout=dropout),
    ])
    return layers


def QuartznetEncoder(feat_in: int = 64, filters: List[int] = [256, 256, 512, 512, 512],
                     kernel_sizes: List[int] = [33, 39, 51, 63, 75], repeat_blocks: int = 1,
                     dropout: float = 0.0) -> nn.Module:
    """QuartzNet 5x5 (repeat_blocks=1) / 15x5 (repeat_blocks=3), reference quartznet/blocks.py:413-434.
    Returns a MultiSequential (same keys "0.mconv.0.conv.weight" ...) that packs a reference-layout input once
    and keeps the activations in the internal bf16 layout through all blocks."""
    return EncoderSequential(stem(feat_in), *body(filters, kernel_sizes, repeat_blocks, dropout))
