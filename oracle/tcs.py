"""Oracle restatement of the QuartzNet / Citrinet time-channel-separable conv stacks.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference:
  src/thunder/quartznet/blocks.py  MaskedConv1d :93-182, _get_conv_bn_layer :185-224,
                                   QuartznetBlock :231-338, stem :341-358, body :361-410,
                                   QuartznetEncoder :413-434
  src/thunder/citrinet/blocks.py   SqueezeExcite :48-83, CitrinetBlock :86-197, stem :200-216,
                                   body :219-255, CitrinetEncoder :258-278
  src/thunder/blocks.py            conv1d_decoder :199-216, linear_decoder :226-248

The oracle does not build nn.Modules: it evaluates an architecture description (`BlockSpec` list)
directly against a reference-layout state dict (same keys the reference's `state_dict()` has), using
torch CPU functional ops in fp32.  `emulate_bf16=True` re-orders the arithmetic the way the HIP path
does (BN folded into the pointwise weights, storage rounded to bf16 at the points the kernels round)
so that the GPU parity tests can use a tight tolerance; `emulate_bf16=False` follows the reference
op order exactly and is what the golden fixtures pin.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .primitives import bf16_round, conv_out_length, lengths_to_mask, same_padding

BN_EPS = 1e-3          # quartznet/blocks.py:222
BN_MOMENTUM = 0.1
SE_REDUCTION = 8       # citrinet/blocks.py:154


@dataclass(frozen=True)
class BlockSpec:
    in_ch: int
    out_ch: int
    repeat: int
    kernel: int
    stride: int = 1
    dilation: int = 1
    residual: bool = True
    separable: bool = True
    family: str = "quartznet"          # "quartznet" | "citrinet" (stride / SE rules differ, A8)

    @property
    def has_se(self) -> bool:
        return self.family == "citrinet"

    def repeat_stride(self, r: int) -> int:
        if self.family == "citrinet":                      # citrinet/blocks.py:128 "only stride the last one"
            return self.stride if r == self.repeat - 1 else 1
        return self.stride                                  # quartznet/blocks.py:266-296

    @property
    def residual_stride(self) -> int:
        if self.stride == 1:
            return 1
        return self.stride if self.family == "citrinet" else self.stride ** self.repeat   # A8

    @property
    def mconv_step(self) -> int:
        """Index distance between consecutive repeats inside `mconv` (conv(s), BN, ReLU, Dropout)."""
        return 5 if self.separable else 4


def quartznet_arch(feat_in: int = 64, filters: Sequence[int] = (256, 256, 512, 512, 512),
                   kernel_sizes: Sequence[int] = (33, 39, 51, 63, 75), repeat_blocks: int = 1) -> List[BlockSpec]:
    """quartznet/blocks.py:341-434.  repeat_blocks=1 -> 5x5, 3 -> 15x5."""
    blocks = [BlockSpec(feat_in, 256, repeat=1, kernel=33, stride=2, residual=False)]
    f_in = 256
    for f, k in zip(filters, kernel_sizes):
        for _ in range(repeat_blocks):
            blocks.append(BlockSpec(f_in, f, repeat=5, kernel=k))
            f_in = f
    blocks.append(BlockSpec(f_in, 512, repeat=1, kernel=87, dilation=2, residual=False))
    blocks.append(BlockSpec(512, 1024, repeat=1, kernel=1, residual=False, separable=False))
    return blocks


def citrinet_arch(filters: Sequence[int], kernel_sizes: Sequence[int], strides: Sequence[int],
                  feat_in: int = 80) -> List[BlockSpec]:
    """citrinet/blocks.py:200-278 (256-channel stem K=5, 640-channel K=41 head hard-coded there)."""
    blocks = [BlockSpec(feat_in, 256, repeat=1, kernel=5, residual=False, family="citrinet")]
    f_in = 256
    for f, k, s in zip(filters, kernel_sizes, strides):
        blocks.append(BlockSpec(f_in, f, repeat=5, kernel=k, stride=s, family="citrinet"))
        f_in = f
    blocks.append(BlockSpec(f_in, 640, repeat=1, kernel=41, residual=False, family="citrinet"))
    return blocks


# ----------------------------------------------------------------------------------------------
# building pieces
# ----------------------------------------------------------------------------------------------

def _mask_time(x: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
    """Zero frames >= length (quartznet/blocks.py:158-167); applied before EVERY conv (A2)."""
    m = lengths_to_mask(lengths, x.shape[-1]).unsqueeze(1)
    return torch.where(m, x, torch.zeros_like(x))


def masked_conv(x, lengths, w, stride, padding, dilation, groups):
    y = F.conv1d(_mask_time(x, lengths), w, None, stride, padding, dilation, groups)
    return y, conv_out_length(lengths, w.shape[-1], stride, padding, dilation)


_CALIB = {"rng": None}      # set only inside synth_encoder_state(calibrate=True)


def batch_norm(y: torch.Tensor, p: Dict[str, torch.Tensor], prefix: str, training: bool,
               new_stats: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """BatchNorm1d(eps=1e-3, momentum=0.1).  Train mode: biased batch statistics over (B, T)
    INCLUDING padded frames (A4); running stats updated with the unbiased variance."""
    g, b = p[prefix + "weight"], p[prefix + "bias"]
    if _CALIB["rng"] is not None:
        # synthetic-weight calibration: make the running stats look like a trained net's (output O(1))
        import numpy as np
        rng = _CALIB["rng"]
        m, v = y.mean(dim=(0, 2)), y.var(dim=(0, 2), unbiased=False) + 1e-6
        c = m.shape[0]
        p[prefix + "running_mean"] = m + v.sqrt() * torch.from_numpy((0.2 * rng.standard_normal(c)).astype(np.float32))
        p[prefix + "running_var"] = v * torch.from_numpy(rng.uniform(0.7, 1.4, c).astype(np.float32))
    if training:
        mean = y.mean(dim=(0, 2))
        var = y.var(dim=(0, 2), unbiased=False)
        if new_stats is not None:
            n = y.shape[0] * y.shape[2]
            new_stats[prefix + "running_mean"] = (1 - BN_MOMENTUM) * p[prefix + "running_mean"] + BN_MOMENTUM * mean.detach()
            new_stats[prefix + "running_var"] = (1 - BN_MOMENTUM) * p[prefix + "running_var"] + BN_MOMENTUM * var.detach() * n / max(n - 1, 1)
    else:
        mean, var = p[prefix + "running_mean"], p[prefix + "running_var"]
    scale = g / torch.sqrt(var + BN_EPS)
    return (y - mean[None, :, None]) * scale[None, :, None] + b[None, :, None]


def bn_fold(p: Dict[str, torch.Tensor], prefix: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """Eval-mode BN as per-channel (scale, shift) (A12)."""
    scale = p[prefix + "weight"] / torch.sqrt(p[prefix + "running_var"] + BN_EPS)
    shift = p[prefix + "bias"] - p[prefix + "running_mean"] * scale
    return scale, shift


def squeeze_excite(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """citrinet/blocks.py:70-83: mean over ALL T' frames (A3), Linear(no bias) -> ReLU -> Linear ->
    sigmoid gate."""
    y = x.mean(dim=-1)                        # [B, C]
    y = torch.relu(y @ w1.t()) @ w2.t()
    return x * torch.sigmoid(y).unsqueeze(-1)


# ----------------------------------------------------------------------------------------------
# block / encoder evaluation
# ----------------------------------------------------------------------------------------------

def block_forward(spec: BlockSpec, sd: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor,
                  lengths: torch.Tensor, training: bool = False, emulate_bf16: bool = False,
                  new_stats: Optional[Dict[str, torch.Tensor]] = None):
    """One QuartznetBlock / CitrinetBlock (quartznet/blocks.py:317-338, citrinet/blocks.py:175-197).

    `sd` holds the block's tensors under `prefix` (e.g. "3.") with the reference key layout."""
    if emulate_bf16 and training:
        raise ValueError("bf16 emulation is defined for eval mode only")
    out, out_len = x, lengths
    step = spec.mconv_step
    for r in range(spec.repeat):
        base = r * step
        s = spec.repeat_stride(r)
        pad = same_padding(spec.kernel, s, spec.dilation)
        last = r == spec.repeat - 1
        if spec.separable:
            dw = sd[f"{prefix}mconv.{base}.conv.weight"]
            pw = sd[f"{prefix}mconv.{base + 1}.conv.weight"]
            bnp = f"{prefix}mconv.{base + 2}.layer.0."
            if emulate_bf16:
                mid, mid_len = masked_conv(out, out_len, bf16_round(dw), s, pad, spec.dilation, dw.shape[0])
                mid = bf16_round(_mask_time(mid, mid_len))
                scale, shift = bn_fold(sd, bnp)
                wf = bf16_round(pw[:, :, 0] * scale[:, None])
                y = torch.einsum("oc,bct->bot", wf, mid) + shift[None, :, None]
                y_len = mid_len
            else:
                mid, mid_len = masked_conv(out, out_len, dw, s, pad, spec.dilation, dw.shape[0])
                y, y_len = masked_conv(mid, mid_len, pw, 1, 0, 1, 1)
                y = batch_norm(y, sd, bnp, training, new_stats)
        else:
            w = sd[f"{prefix}mconv.{base}.conv.weight"]
            bnp = f"{prefix}mconv.{base + 1}.layer.0."
            if emulate_bf16:
                scale, shift = bn_fold(sd, bnp)
                wf = bf16_round(w * scale[:, None, None])
                y, y_len = masked_conv(out, out_len, wf, s, pad, spec.dilation, 1)
                y = y + shift[None, :, None]
            else:
                y, y_len = masked_conv(out, out_len, w, s, pad, spec.dilation, 1)
                y = batch_norm(y, sd, bnp, training, new_stats)
        if not last:
            y = torch.relu(y)
            if emulate_bf16:
                y = bf16_round(y)
        out, out_len = y, y_len

    if spec.has_se:
        se_idx = (spec.repeat - 1) * step + (3 if spec.separable else 2)
        sep = f"{prefix}mconv.{se_idx}.layer.0.fc."
        if emulate_bf16:
            # HIP path: the last sub-block's (pre-SE) output is stored bf16, pooled in fp32
            out = bf16_round(out)
        out = squeeze_excite(out, sd[sep + "0.weight"], sd[sep + "2.weight"])

    if spec.residual:
        rw = sd[f"{prefix}res.0.conv.weight"]
        rbn = f"{prefix}res.1.layer.0."
        if emulate_bf16:
            scale, shift = bn_fold(sd, rbn)
            wf = bf16_round(rw * scale[:, None, None])
            res, _ = masked_conv(x, lengths, wf, spec.residual_stride, 0, 1, 1)
            res = res + shift[None, :, None]
            if spec.has_se:
                res = bf16_round(res)             # HIP path: the residual branch of an SE block is its own launch, stored bf16
        else:
            res, _ = masked_conv(x, lengths, rw, spec.residual_stride, 0, 1, 1)
            res = batch_norm(res, sd, rbn, training, new_stats)
        out = out + res
    out = torch.relu(out)
    if emulate_bf16:
        out = bf16_round(out)
    return out, out_len


def encoder_forward(arch: List[BlockSpec], sd: Dict[str, torch.Tensor], x: torch.Tensor, lengths: torch.Tensor,
                    training: bool = False, emulate_bf16: bool = False, prefix: str = "",
                    new_stats: Optional[Dict[str, torch.Tensor]] = None):
    """MultiSequential of blocks (quartznet/blocks.py:431-434, citrinet/blocks.py:275-278)."""
    if emulate_bf16:
        x = bf16_round(x)
    for i, spec in enumerate(arch):
        x, lengths = block_forward(spec, sd, f"{prefix}{i}.", x, lengths, training, emulate_bf16, new_stats)
    return x, lengths


def conv1d_decoder_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, prefix: str = "",
                           emulate_bf16: bool = False) -> torch.Tensor:
    """1x1 conv + bias -> [B, V, T'] logits (blocks.py:199-216)."""
    w, b = sd[prefix + "weight"], sd[prefix + "bias"]
    if emulate_bf16:
        w = bf16_round(w)
    return torch.einsum("oc,bct->bot", w[:, :, 0], x) + b[None, :, None]


def linear_decoder_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, prefix: str = "",
                           emulate_bf16: bool = False) -> torch.Tensor:
    """transpose -> dropout(eval: id) -> Linear -> transpose (blocks.py:226-248); keys "2.weight"/"2.bias"."""
    w, b = sd[prefix + "2.weight"], sd[prefix + "2.bias"]
    if emulate_bf16:
        w = bf16_round(w)
    return torch.einsum("oc,bct->bot", w, x) + b[None, :, None]


# ----------------------------------------------------------------------------------------------
# synthetic, platform-stable weights with the reference's key layout (SURVEY 8d "Synthetic inputs")
# ----------------------------------------------------------------------------------------------

def _xavier_uniform(rng, shape) -> torch.Tensor:
    import numpy as np
    receptive = 1
    for s in shape[2:]:
        receptive *= s
    fan_in, fan_out = shape[1] * receptive, shape[0] * receptive
    bound = (6.0 / (fan_in + fan_out)) ** 0.5
    return torch.from_numpy(rng.uniform(-bound, bound, size=shape).astype(np.float32))


def synth_encoder_state(arch: List[BlockSpec], seed: int = 0, gain: float = 1.0,
                        calibrate: bool = False, main_gamma: float = 1.0) -> Dict[str, torch.Tensor]:
    """Deterministic (numpy PCG64) weights: xavier-uniform convs (the reference's default init,
    quartznet/blocks.py:59-90), BN affine ~ (1 +- 0.1, +-0.1), running_mean ~ N(0, 0.1),
    running_var ~ U(0.5, 1.5) so that BN folding is exercised.

    calibrate=True additionally replaces every BN's running statistics by (perturbed) statistics of
    that layer's pre-BN activations on a seeded calibration batch, as training would, so activations
    stay O(1) through all 18 blocks of QuartzNet15x5 instead of decaying to the BN bias; parity tests
    and the benchmark then exercise the whole stack with realistic magnitudes."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, torch.Tensor] = {}

    def bn(prefix, c):
        sd[prefix + "weight"] = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(c)).astype(np.float32))
        sd[prefix + "bias"] = torch.from_numpy((0.1 * rng.standard_normal(c)).astype(np.float32))
        sd[prefix + "running_mean"] = torch.from_numpy((0.1 * rng.standard_normal(c)).astype(np.float32))
        sd[prefix + "running_var"] = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32))
        sd[prefix + "num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    for i, spec in enumerate(arch):
        p = f"{i}."
        cin = spec.in_ch
        for r in range(spec.repeat):
            base = r * spec.mconv_step
            if spec.separable:
                # depthwise gain: keep the activation scale O(1) through the stack
                sd[f"{p}mconv.{base}.conv.weight"] = _xavier_uniform(rng, (cin, 1, spec.kernel)) * gain
                sd[f"{p}mconv.{base + 1}.conv.weight"] = _xavier_uniform(rng, (spec.out_ch, cin, 1)) * gain
                bnp = f"{p}mconv.{base + 2}.layer.0."
            else:
                sd[f"{p}mconv.{base}.conv.weight"] = _xavier_uniform(rng, (spec.out_ch, cin, spec.kernel)) * gain
                bnp = f"{p}mconv.{base + 1}.layer.0."
            bn(bnp, spec.out_ch)
            if spec.residual and r == spec.repeat - 1:
                # trained residual nets keep the skip path dominant; a random 5-layer main branch at full weight
                # makes the stack chaotic (bf16 rounding noise then grows ~1.3x per block)
                sd[bnp + "weight"] = sd[bnp + "weight"] * main_gamma
            cin = spec.out_ch
        if spec.has_se:
            se_idx = (spec.repeat - 1) * spec.mconv_step + (3 if spec.separable else 2)
            c = spec.out_ch
            sd[f"{p}mconv.{se_idx}.layer.0.fc.0.weight"] = _xavier_uniform(rng, (c // SE_REDUCTION, c))
            sd[f"{p}mconv.{se_idx}.layer.0.fc.2.weight"] = _xavier_uniform(rng, (c, c // SE_REDUCTION))
        if spec.residual:
            sd[f"{p}res.0.conv.weight"] = _xavier_uniform(rng, (spec.out_ch, spec.in_ch, 1)) * gain
            bn(f"{p}res.1.layer.0.", spec.out_ch)
    if calibrate:
        t = 640
        x = torch.from_numpy(rng.standard_normal((2, arch[0].in_ch, t)).astype(np.float32))
        lengths = torch.tensor([t, (7 * t) // 8])
        _CALIB["rng"] = rng
        try:
            with torch.no_grad():
                encoder_forward(arch, sd, x, lengths)
        finally:
            _CALIB["rng"] = None
    return sd


def synth_decoder_state(in_ch: int, num_classes: int, seed: int = 1, linear: bool = False, gain: float = 1.0):
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    if linear:
        return {"2.weight": _xavier_uniform(rng, (num_classes, in_ch)) * gain,
                "2.bias": torch.from_numpy((0.1 * rng.standard_normal(num_classes)).astype(np.float32))}
    return {"weight": _xavier_uniform(rng, (num_classes, in_ch, 1)) * gain,
            "bias": torch.from_numpy((0.1 * rng.standard_normal(num_classes)).astype(np.float32))}
