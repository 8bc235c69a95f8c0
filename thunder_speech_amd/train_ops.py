"""Training-mode encoder ops (fine-tuning with the encoder unfrozen): one torch.autograd.Function per reference op, each
forward AND backward a call into csrc/train_enc.hip (fp32 activations, reference layout [B, C, T]).  PyTorch only chains the
Functions; no ATen compute op touches an activation."""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib


# Operand precision of the pointwise-conv GEMMs (forward and both backward products): "fp32" (default: the reference's
# arithmetic) or "bf16" (opt-in mixed precision: bf16 operands, fp32 accumulation / results / master weights, like Lightning's
# precision="bf16-mixed" for the reference).  Set through `set_gemm_precision`.
_GEMM_BF16 = False


def set_gemm_precision(precision: str) -> None:
    global _GEMM_BF16
    if precision not in ("fp32", "bf16"):
        raise ValueError(f"precision must be 'fp32' or 'bf16', got {precision!r}")
    _GEMM_BF16 = precision == "bf16"


def _bf16(t: Tensor) -> Tensor:
    """bf16 operand copy of a contiguous fp32 tensor (ts_train_cast_bf16)."""
    y = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    _lib.check(_lib.lib().ts_train_cast_bf16(t.data_ptr(), y.data_ptr(), t.numel(), _s(t)), "ts_train_cast_bf16")
    return y


def _s(t: Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def _f32(t: Tensor) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError("thunder_speech_amd training ops run on the GPU only (no CPU fallback)")
    return t.to(torch.float32).contiguous()


class DepthwiseConv(torch.autograd.Function):
    """MaskedConv1d with groups = C: y = conv(mask(x, len_in)); backward masks dx the same way.  With `len_out` the output is
    zeroed from len_out[b] on (the re-masking the following MaskedConv1d would apply) and so is the incoming gradient."""

    @staticmethod
    def forward(ctx, x, w, len_in, k, stride, dil, pad, len_out=None):
        x, w2 = _f32(x), _f32(w).view(w.shape[0], -1)
        b, c, t_in = x.shape
        t_out = (t_in + 2 * pad - dil * (k - 1) - 1) // stride + 1
        y = torch.empty(b, c, t_out, dtype=torch.float32, device=x.device)
        st = _lib.lib().ts_train_dwconv_fwd(x.data_ptr(), len_in.data_ptr(), len_out.data_ptr() if len_out is not None else None,
                                            w2.data_ptr(), y.data_ptr(), b, c, t_in, t_out, k, stride, dil, pad, _s(x))
        _lib.check(st, "ts_train_dwconv_fwd")
        ctx.save_for_backward(x, w2, len_in)
        ctx.len_out = len_out
        ctx.geom = (k, stride, dil, pad, t_out, w.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2, len_in = ctx.saved_tensors
        k, stride, dil, pad, t_out, wshape = ctx.geom
        dy = _f32(dy)
        b, c, t_in = x.shape
        dx, dw = torch.empty_like(x), torch.empty_like(w2)
        lo = ctx.len_out
        st = _lib.lib().ts_train_dwconv_bwd(dy.data_ptr(), x.data_ptr(), len_in.data_ptr(), lo.data_ptr() if lo is not None else None,
                                            w2.data_ptr(), dx.data_ptr(), dw.data_ptr(), b, c, t_in, t_out, k, stride, dil, pad, _s(x))
        _lib.check(st, "ts_train_dwconv_bwd")
        return dx, dw.view(wshape), None, None, None, None, None, None


class MaskTime(torch.autograd.Function):
    """Zero the frames >= length (the re-masking in front of every MaskedConv1d); the gradient is masked the same way."""

    @staticmethod
    def forward(ctx, x, lens):
        x = _f32(x)
        b, c, t = x.shape
        y = torch.empty_like(x)
        _lib.check(_lib.lib().ts_train_mask_time(x.data_ptr(), lens.data_ptr(), y.data_ptr(), b, c, t, _s(x)), "ts_train_mask_time")
        ctx.save_for_backward(lens)
        return y

    @staticmethod
    def backward(ctx, dy):
        (lens,) = ctx.saved_tensors
        dy = _f32(dy)
        b, c, t = dy.shape
        dx = torch.empty_like(dy)
        _lib.check(_lib.lib().ts_train_mask_time(dy.data_ptr(), lens.data_ptr(), dx.data_ptr(), b, c, t, _s(dy)), "ts_train_mask_time")
        return dx, None


class PointwiseConv(torch.autograd.Function):
    """1x1 conv without bias on an already masked input: v[b] = W . u[b] (rocBLAS), du = W^T dv, dW = sum_b dv u^T."""

    @staticmethod
    def forward(ctx, u, w):
        u, w2 = _f32(u), _f32(w).view(w.shape[0], -1)
        b, c_in, t = u.shape
        c_out = w2.shape[0]
        v = torch.empty(b, c_out, t, dtype=torch.float32, device=u.device)
        prec = 1 if _GEMM_BF16 else 0
        if prec:
            u, w2 = _bf16(u), _bf16(w2)            # the bf16 copies are what the backward needs too: half the saved bytes
        _lib.check(_lib.lib().ts_train_pwconv_fwd(u.data_ptr(), w2.data_ptr(), v.data_ptr(), b, c_in, c_out, t, prec, _s(v)),
                   "ts_train_pwconv_fwd")
        ctx.save_for_backward(u, w2)
        ctx.wshape, ctx.prec = w.shape, prec
        return v

    @staticmethod
    def backward(ctx, dv):
        u, w2 = ctx.saved_tensors
        dv = _f32(dv)
        b, c_in, t = u.shape
        c_out = w2.shape[0]
        du = torch.empty(u.shape, dtype=torch.float32, device=u.device)
        dw = torch.empty(w2.shape, dtype=torch.float32, device=u.device)
        ws = torch.empty(b * c_out * c_in, dtype=torch.float32, device=u.device)
        if ctx.prec:
            dv = _bf16(dv)
        st = _lib.lib().ts_train_pwconv_bwd(dv.data_ptr(), u.data_ptr(), w2.data_ptr(), du.data_ptr(), dw.data_ptr(), ws.data_ptr(), b, c_in,
                                            c_out, t, ctx.prec, _s(du))
        _lib.check(st, "ts_train_pwconv_bwd")
        return du, dw.view(ctx.wshape)


class BatchNormTrain(torch.autograd.Function):
    """BatchNorm1d in train mode (+ optional ReLU): statistics over all B*T frames (quirk A4).  `running` = (running_mean,
    running_var, momentum, num_batches_tracked) or None: the module's running statistics, updated by the same launch."""

    @staticmethod
    def forward(ctx, v, gamma, beta, eps, relu, running):
        v, g, be = _f32(v), _f32(gamma), _f32(beta)
        b, c, t = v.shape
        y = torch.empty_like(v)
        mr = torch.empty(c, 2, dtype=torch.float32, device=v.device)
        ws = torch.empty(16 * c, dtype=torch.float64, device=v.device)
        rm, rv, mom, nbt = running if running is not None else (None, None, 0.0, None)
        st = _lib.lib().ts_train_bn_fwd(v.data_ptr(), g.data_ptr(), be.data_ptr(), y.data_ptr(), mr.data_ptr(), ws.data_ptr(), b, c, t,
                                        float(eps), int(relu), rm.data_ptr() if rm is not None else None,
                                        rv.data_ptr() if rv is not None else None, float(mom),
                                        nbt.data_ptr() if nbt is not None else None, _s(v))
        _lib.check(st, "ts_train_bn_fwd")
        ctx.save_for_backward(v, y, g, mr)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        v, y, g, mr = ctx.saved_tensors
        dy = _f32(dy)
        b, c, t = v.shape
        dv = torch.empty_like(v)
        dg, db = torch.empty(c, dtype=torch.float32, device=v.device), torch.empty(c, dtype=torch.float32, device=v.device)
        ws = torch.empty(16 * c, dtype=torch.float64, device=v.device)
        st = _lib.lib().ts_train_bn_bwd(dy.data_ptr(), y.data_ptr(), v.data_ptr(), g.data_ptr(), mr.data_ptr(), dv.data_ptr(), dg.data_ptr(),
                                        db.data_ptr(), ws.data_ptr(), b, c, t, int(ctx.relu), _s(v))
        _lib.check(st, "ts_train_bn_bwd")
        return dv, dg, db, None, None, None


class AddRelu(torch.autograd.Function):
    """out = relu(a + b) (b may be None)."""

    @staticmethod
    def forward(ctx, a, b):
        a = _f32(a)
        bb = _f32(b) if b is not None else None
        out = torch.empty_like(a)
        st = _lib.lib().ts_train_add_relu_fwd(a.data_ptr(), bb.data_ptr() if bb is not None else None, out.data_ptr(), a.numel(), _s(a))
        _lib.check(st, "ts_train_add_relu_fwd")
        ctx.save_for_backward(out)
        ctx.has_b = b is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        dout = _f32(dout)
        din = torch.empty_like(out)
        _lib.check(_lib.lib().ts_train_relu_bwd(dout.data_ptr(), out.data_ptr(), din.data_ptr(), out.numel(), _s(out)), "ts_train_relu_bwd")
        return din, (din if ctx.has_b else None)


class Dropout(torch.autograd.Function):
    """nn.Dropout in train mode: y = x * keep / (1 - p), keep ~ Bernoulli(1 - p) drawn per element from a Philox stream
    (csrc/augment.hip).  Nothing is saved: the backward pass re-draws the mask from the same seed."""

    @staticmethod
    def forward(ctx, x, p, seed):
        x = _f32(x)
        y = torch.empty_like(x)
        _lib.check(_lib.lib().ts_dropout(x.data_ptr(), y.data_ptr(), x.numel(), float(p), int(seed), _s(x)), "ts_dropout")
        ctx.p, ctx.seed = float(p), int(seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _f32(dy)
        dx = torch.empty_like(dy)
        _lib.check(_lib.lib().ts_dropout(dy.data_ptr(), dx.data_ptr(), dy.numel(), ctx.p, ctx.seed, _s(dy)), "ts_dropout")
        return dx, None, None


def dropout(x: Tensor, p: float, training: bool) -> Tensor:
    if not training or p <= 0.0:
        return x
    from .rng import next_seed
    return Dropout.apply(x, p, next_seed())


class SubsampleMask(torch.autograd.Function):
    """Input side of a strided 1x1 MaskedConv1d: zero the frames >= length, keep every `stride`-th frame."""

    @staticmethod
    def forward(ctx, x, lens, stride, t_out):
        x = _f32(x)
        b, c, t_in = x.shape
        y = torch.empty(b, c, t_out, dtype=torch.float32, device=x.device)
        st = _lib.lib().ts_train_subsample_mask(x.data_ptr(), lens.data_ptr(), y.data_ptr(), b, c, t_in, t_out, stride, 0, _s(x))
        _lib.check(st, "ts_train_subsample_mask")
        ctx.save_for_backward(lens)
        ctx.geom = (b, c, t_in, t_out, stride)
        return y

    @staticmethod
    def backward(ctx, dy):
        (lens,) = ctx.saved_tensors
        b, c, t_in, t_out, stride = ctx.geom
        dy = _f32(dy)
        dx = torch.empty(b, c, t_in, dtype=torch.float32, device=dy.device)
        st = _lib.lib().ts_train_subsample_mask(dy.data_ptr(), lens.data_ptr(), dx.data_ptr(), b, c, t_in, t_out, stride, 1, _s(dy))
        _lib.check(st, "ts_train_subsample_mask")
        return dx, None, None, None


class SqueezeExciteTrain(torch.autograd.Function):
    """SqueezeExcite.forward with autograd (citrinet/blocks.py:70-83): y = x * sigmoid(W2 relu(W1 mean_t(x))), the mean over ALL
    frames (quirk A3).  The passes over the activation are HIP launches (csrc/train_extra.hip); the [B, C] bottleneck is four
    tiny GEMMs."""

    @staticmethod
    def forward(ctx, x, w1, w2):
        x, w1, w2 = _f32(x), _f32(w1), _f32(w2)
        b, c, t = x.shape
        L = _lib.lib()
        mean = torch.empty(b, c, dtype=torch.float32, device=x.device)
        _lib.check(L.ts_train_se_pool(x.data_ptr(), mean.data_ptr(), b * c, t, _s(x)), "ts_train_se_pool")
        h = torch.relu(mean @ w1.t())
        g = torch.sigmoid(h @ w2.t()).contiguous()
        y = torch.empty_like(x)
        _lib.check(L.ts_train_se_scale(x.data_ptr(), g.data_ptr(), None, y.data_ptr(), b * c, t, _s(x)), "ts_train_se_scale")
        ctx.save_for_backward(x, w1, w2, mean, h, g)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, mean, h, g = ctx.saved_tensors
        dy = _f32(dy)
        b, c, t = x.shape
        L = _lib.lib()
        dg = torch.empty(b, c, dtype=torch.float32, device=x.device)
        _lib.check(L.ts_train_se_rowdot(dy.data_ptr(), x.data_ptr(), dg.data_ptr(), b * c, t, _s(x)), "ts_train_se_rowdot")
        dz = dg * g * (1.0 - g)
        dw2 = dz.t() @ h
        dh = (dz @ w2) * (h > 0).to(dz.dtype)
        dw1 = dh.t() @ mean
        dmean = (dh @ w1).contiguous()
        dx = torch.empty_like(x)
        _lib.check(L.ts_train_se_scale(dy.data_ptr(), g.data_ptr(), dmean.data_ptr(), dx.data_ptr(), b * c, t, _s(x)), "ts_train_se_scale")
        return dx, dw1, dw2


def batch_norm_train(bn: torch.nn.BatchNorm1d, v: Tensor, relu: bool) -> Tensor:
    """BatchNorm1d(train) through the kernels + the module's running-statistics update (momentum, unbiased variance)."""
    running = None
    if bn.track_running_stats and bn.running_mean is not None:
        if bn.running_mean.dtype != torch.float32 or bn.running_var.dtype != torch.float32 or not bn.running_mean.is_cuda:
            raise RuntimeError("batch_norm_train: fp32 running statistics on the GPU only")
        # momentum=None (cumulative average) needs the counter's value: one host read, the reference default is 0.1
        m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
        running = (bn.running_mean, bn.running_var, m, bn.num_batches_tracked)
    y = BatchNormTrain.apply(v, bn.weight, bn.bias, bn.eps, relu, running)
    if running is not None:
        # the launch updated the buffers through raw pointers: make the change visible to `_version`-keyed caches
        # (blocks._PackedCache folds running_mean / running_var into the inference weights)
        for t in (bn.running_mean, bn.running_var, bn.num_batches_tracked):
            torch.autograd.graph.increment_version(t)
    return y
