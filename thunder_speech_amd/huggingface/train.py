"""wav2vec2 fine-tuning on the HIP kernels: the training-mode forward of `transformers.Wav2Vec2Model` as a chain of autograd nodes whose
forward and backward are C-ABI launches (csrc/gemm_f32.hip for every product, csrc/w2v_train.hip for the rest).

The reference fine-tunes HuggingFace CTC checkpoints through `BaseCTCModule.training_step` (src/thunder/module.py:102-127) with the conv feature
extractor frozen (`_HuggingFaceEncoderAdapt.__init__`, src/thunder/huggingface/compatibility.py:27-28 -> `freeze_feature_encoder()`); autograd
then runs through feature_projection, the train-time time masking (`mask_time_prob`), the weight-normalised positional conv and the transformer
layers with their dropouts and LayerDrop (transformers' modeling_wav2vec2.py: Wav2Vec2Model.forward, Wav2Vec2Encoder /
Wav2Vec2EncoderStableLayerNorm; forward restated for eval mode in oracle/w2v.py).  This module restates that training-mode forward on
time-major f32 activations [B, T, C] (the reference's arithmetic) and owns the backward: PyTorch only chains the nodes and accumulates the
parameters' .grad; no ATen kernel touches an activation.  The parameters stay the transformers module's own (`original_encoder`, reference
state-dict keys), so optimizers and checkpoints see no difference.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
from torch import Tensor

from .. import _lib
from ..rng import next_seed


def _s(t: Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _f32c(t: Tensor) -> Tensor:
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()


def _gemm(a, a_rs, a_cs, b, b_rs, b_cs, c, ldc, m, n, k, *, sa=0, sa2=0, ska=0, sb=0, sb2=0, skb=0, sc=0, sc2=0, bias=None, nkb=1, batch=1,
          batch2=1, beta=False, a_off=0, b_off=0, c_off=0):
    """C (+)= A . B on ts_gemm_f32_b2 (element strides; *_off = element offsets into the tensors)."""
    st = _lib.lib().ts_gemm_f32_b2(a.data_ptr() + 4 * a_off, a_rs, a_cs, sa, sa2, ska, b.data_ptr() + 4 * b_off, b_rs, b_cs, sb, sb2, skb,
                                   c.data_ptr() + 4 * c_off, ldc, sc, sc2, bias.data_ptr() if bias is not None else None, m, n, k, nkb, batch, batch2,
                                   int(beta), _s(c))
    _lib.check(st, "ts_gemm_f32_b2")


def _colsum(x: Tensor, rows: int, c: int) -> Tensor:
    out = torch.zeros(c, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ts_w2v_colsum(x.data_ptr(), rows, c, c, out.data_ptr(), _s(x)), "ts_w2v_colsum")
    return out


class Linear(torch.autograd.Function):
    """y = x W^T + b over the last axis (nn.Linear)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = _f32c(x), _f32c(w)
        n, k = w.shape
        rows = x.numel() // k
        y = torch.empty(*x.shape[:-1], n, dtype=torch.float32, device=x.device)
        _gemm(x, k, 1, w, 1, k, y, n, rows, n, k, bias=_f32c(b) if b is not None else None)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _f32c(dy)
        n, k = w.shape
        rows = x.numel() // k
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _gemm(dy, n, 1, w, k, 1, dx, k, rows, k, n)                     # dx[r][k] = sum_n dy[r][n] W[n][k]
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            _gemm(dy, 1, n, x, k, 1, dw, k, n, k, rows)                     # dW[n][k] = sum_r dy[r][n] x[r][k]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy, rows, n)
        return dx, dw, db


def _pad32(n: int) -> int:
    """Padded length of a contraction index: a multiple of 32 (ts_gemm_nt_bf16), of 256 for long ones so that split-K has powers of two to choose from."""
    return (n + 255) // 256 * 256 if n > 1024 else (n + 31) // 32 * 32


def _cast(x2: Tensor, rows: int, c: int, plain: bool, transposed: bool, colsum: Optional[Tensor] = None):
    """(bf16 copy [rows][c] or None, transposed bf16 copy [c][pad32(rows)] or None) of the f32 matrix x2 (contiguous [rows][c]): ONE launch, which also
    adds the column sums of x2 to `colsum` (f32 [c]) when given -- a linear layer's bias gradient out of the pass that casts dy."""
    dev = x2.device
    y = torch.empty(rows, c, dtype=torch.bfloat16, device=dev) if plain else None
    rp = _pad32(rows)
    yt = torch.empty(c, rp, dtype=torch.bfloat16, device=dev) if transposed else None
    st = _lib.lib().ts_w2v_cast_bf16_t_colsum(x2.data_ptr(), c, rows, c, y.data_ptr() if plain else None, c, yt.data_ptr() if transposed else None, rp, rp,
                                              colsum.data_ptr() if colsum is not None else None, _s(x2))
    _lib.check(st, "ts_w2v_cast_bf16_t_colsum")
    return y, yt


def _gemm_nt(a16: Tensor, lda: int, w16: Tensor, ldw: int, out: Tensor, rows: int, n: int, k: int, bias: Optional[Tensor] = None):
    """out[r][j] = sum_i a16[r][i] w16[j][i] (+ bias[j]) on the bf16 matrix-core GEMM (csrc/gemm_nt.hip), f32 accumulation and result."""
    st = _lib.lib().ts_gemm_nt_bf16(a16.data_ptr(), lda, w16.data_ptr(), ldw, bias.data_ptr() if bias is not None else None, None, 0, out.data_ptr(), n,
                                    None, 0, rows, n, k, 0, _s(out))
    _lib.check(st, "ts_gemm_nt_bf16")


SPLITK_TARGET_WGS = 192     # split-K while a product has fewer workgroups than this (256 x 256 tiles, one per CU; same-box sweep of the 8 x 10 s step,
                            # tools/diag/splitk_ab.py: 128 -> 30.0 ms, 192 -> 28.9, 224 -> 29.8, 256 -> 29.6)


def _gemm_nt_splitk(a16: Tensor, lda: int, w16: Tensor, ldw: int, out: Tensor, rows: int, n: int, k: int, bias: Optional[Tensor] = None, min_k: int = 0):
    """The same product for FEW output tiles and a LONG contraction (the weight gradients: 1024 x 1024 outputs are 16 tiles of 256 x 256 on 256 compute
    units): split-K into f32 partials (ts_gemm_nt_bf16_splitk) + an ordered sum (ts_w2v_sum_parts), splits chosen for ~128 workgroups."""
    tiles = ((rows + 255) // 256) * ((n + 255) // 256)
    splits = 1
    while tiles * splits < SPLITK_TARGET_WGS and k % (64 * splits) == 0 and k // (2 * splits) >= 256:
        splits *= 2
    if splits == 1 or (rows * n) % 4 or k < min_k or (bias is not None and n % 4):
        return _gemm_nt(a16, lda, w16, ldw, out, rows, n, k, bias)
    parts = torch.empty(splits, rows, n, dtype=torch.float32, device=out.device)
    L = _lib.lib()
    _lib.check(L.ts_gemm_nt_bf16_splitk(a16.data_ptr(), lda, w16.data_ptr(), ldw, parts.data_ptr(), rows, n, k, splits, _s(out)), "ts_gemm_nt_bf16_splitk")
    if bias is not None:
        _lib.check(L.ts_w2v_sum_parts_bias(parts.data_ptr(), bias.data_ptr(), n, out.data_ptr(), rows * n, splits, _s(out)), "ts_w2v_sum_parts_bias")
    else:
        _lib.check(L.ts_w2v_sum_parts(parts.data_ptr(), out.data_ptr(), rows * n, splits, _s(out)), "ts_w2v_sum_parts")


class LinearMixed(torch.autograd.Function):
    """nn.Linear in mixed precision (what Lightning's precision="bf16-mixed" gives the reference's training_step, module.py:102-127): bf16 operands,
    f32 accumulation, f32 activations / master weights / gradients.  All three products are the NT form of csrc/gemm_nt.hip:
        y  = x16 . w16^T                    x16 [M][K], w16 [N][K]
        dx = dy16 . (w16^T)^T               dy16 [M][N], wT16 [K][N]
        dW = dyT16 . (xT16)^T               dyT16 [N][Mp], xT16 [K][Mp]   (Mp = M padded to 32 with zeros: the contraction index)
    The transposed copies come out of the same cast launch as the plain ones; the node keeps xT16 (half the bytes of the f32 input) and wT16."""

    @staticmethod
    def supported(n: int, k: int) -> bool:
        return n % 32 == 0 and k % 32 == 0

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = _f32c(x), _f32c(w)
        n, k = w.shape
        rows = x.numel() // k
        x16, xt16 = _cast(x.view(rows, k), rows, k, True, ctx.needs_input_grad[1])
        w16, wt16 = _cast(w, n, k, True, ctx.needs_input_grad[0])
        y = torch.empty(*x.shape[:-1], n, dtype=torch.float32, device=x.device)
        # few output tiles and a long contraction (the second feed-forward linear: 64 tiles, K = 4096): split-K fills the chip
        _gemm_nt_splitk(x16, k, w16, k, y, rows, n, k, bias=_f32c(b) if b is not None else None, min_k=2048)
        ctx.save_for_backward(xt16, wt16)
        ctx.has_bias, ctx.geom = b is not None, (rows, n, k, x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        xt16, wt16 = ctx.saved_tensors
        rows, n, k, xshape = ctx.geom
        dy = _f32c(dy)
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        fused_db = want_db and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1])       # the bias gradient rides in the cast launch when there is one
        if fused_db:
            db = torch.zeros(n, dtype=torch.float32, device=dy.device)
        dy16, dyt16 = _cast(dy.view(rows, n), rows, n, ctx.needs_input_grad[0], ctx.needs_input_grad[1], colsum=db if fused_db else None)
        if ctx.needs_input_grad[0]:
            dx = torch.empty(xshape, dtype=torch.float32, device=dy.device)
            _gemm_nt_splitk(dy16, n, wt16, _pad32(n), dx, rows, k, n)             # (n % 32 == 0: wT16's pitch is n); split-K when the output has < 128 tiles
        if ctx.needs_input_grad[1]:
            dw = torch.empty(n, k, dtype=torch.float32, device=dy.device)
            rp = _pad32(rows)
            _gemm_nt_splitk(dyt16, rp, xt16, rp, dw, n, k, rp)
        if want_db and not fused_db:
            db = _colsum(dy, rows, n)
        return dx, dw, db


class FFNActLinear(torch.autograd.Function):
    """output_dense(dropout(gelu(z + b1))) of a wav2vec2 feed-forward block in mixed precision, as ONE node: the activation goes straight from z to the bf16
    operands of the second linear (ts_w2v_ffn_act_cast: the f32 activation is never stored), and the backward runs the data-gradient product into a buffer that
    the fused dropout-backward x gelu' launch (ts_w2v_ffn_act_bwd) turns into dz.  Same products, split-K rules and Philox mask as BiasGelu -> Dropout -> LinearMixed."""

    @staticmethod
    def supported(c_mid: int, n: int) -> bool:
        return c_mid % 32 == 0 and n % 32 == 0

    @staticmethod
    def forward(ctx, z, b1, w2, b2, p_drop, seed):
        z, w2 = _f32c(z), _f32c(w2)
        n, k = w2.shape                                   # k = the intermediate width
        rows = z.numel() // k
        L = _lib.lib()
        rp = _pad32(rows)
        x16 = torch.empty(rows, k, dtype=torch.bfloat16, device=z.device)
        xt16 = torch.empty(k, rp, dtype=torch.bfloat16, device=z.device) if ctx.needs_input_grad[2] else None
        _lib.check(L.ts_w2v_ffn_act_cast(z.data_ptr(), b1.data_ptr() if b1 is not None else None, rows, k, float(p_drop), int(seed), x16.data_ptr(),
                                         xt16.data_ptr() if xt16 is not None else None, rp, rp, _s(z)), "ts_w2v_ffn_act_cast")
        w16, wt16 = _cast(w2, n, k, True, ctx.needs_input_grad[0])
        y = torch.empty(*z.shape[:-1], n, dtype=torch.float32, device=z.device)
        _gemm_nt_splitk(x16, k, w16, k, y, rows, n, k, bias=_f32c(b2) if b2 is not None else None, min_k=2048)
        ctx.save_for_backward(z, b1, xt16, wt16)
        ctx.geom = (rows, n, k, z.shape, float(p_drop), int(seed), b2 is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, b1, xt16, wt16 = ctx.saved_tensors
        rows, n, k, zshape, p_drop, seed, has_b2 = ctx.geom
        dy = _f32c(dy)
        L = _lib.lib()
        dz = db1 = dw = db2 = None
        want_db2 = has_b2 and ctx.needs_input_grad[3]
        if want_db2:
            db2 = torch.zeros(n, dtype=torch.float32, device=dy.device)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            dy16, dyt16 = _cast(dy.view(rows, n), rows, n, ctx.needs_input_grad[0], ctx.needs_input_grad[2], colsum=db2)
        elif want_db2:
            db2 = _colsum(dy, rows, n)
        if ctx.needs_input_grad[0]:
            da = torch.empty(rows, k, dtype=torch.float32, device=dy.device)
            _gemm_nt_splitk(dy16, n, wt16, _pad32(n), da, rows, k, n)
            dz = torch.empty(zshape, dtype=torch.float32, device=dy.device)
            _lib.check(L.ts_w2v_ffn_act_bwd(z.data_ptr(), b1.data_ptr() if b1 is not None else None, k, da.data_ptr(), p_drop, seed, dz.data_ptr(), dz.numel(), _s(dy)),
                       "ts_w2v_ffn_act_bwd")
            if b1 is not None and ctx.needs_input_grad[1]:
                db1 = _colsum(dz, rows, k)
        if ctx.needs_input_grad[2]:
            dw = torch.empty(n, k, dtype=torch.float32, device=dy.device)
            rp = _pad32(rows)
            _gemm_nt_splitk(dyt16, rp, xt16, rp, dw, n, k, rp)
        return dz, db1, dw, db2, None, None


FUSED_FFN_ACT = True        # mixed mode: FFNActLinear for gelu -> dropout -> output_dense (False: the three separate nodes, for A/B)
_MIXED = False


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """nn.Linear of the training path in the current precision mode (`train_forward(..., mixed=...)`)."""
    if _MIXED and LinearMixed.supported(w.shape[0], w.shape[1]):
        return LinearMixed.apply(x, w, b)
    return Linear.apply(x, w, b)


class LayerNorm(torch.autograd.Function):
    """y = LayerNorm(x (+ res)) over the last axis; the residual add rides in the launch (both inputs receive the same gradient)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps):
        x = _f32c(x)
        res = _f32c(res) if res is not None else None
        c = x.shape[-1]
        rows = x.numel() // c
        y = torch.empty_like(x)
        st = _lib.lib().ts_w2v_layernorm_fwd(x.data_ptr(), res.data_ptr() if res is not None else None, None, gamma.data_ptr(), beta.data_ptr(), eps,
                                             rows, c, 0, y.data_ptr(), None, _s(x))
        _lib.check(st, "ts_w2v_layernorm_fwd")
        ctx.save_for_backward(x, res, gamma)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, res, gamma = ctx.saved_tensors
        dy = _f32c(dy)
        c = x.shape[-1]
        rows = x.numel() // c
        L = _lib.lib()
        dx = torch.empty_like(x)
        dg = torch.empty(c, dtype=torch.float32, device=x.device)
        db = torch.empty(c, dtype=torch.float32, device=x.device)
        ws = torch.empty(L.ts_w2v_layernorm_bwd_workspace(rows, c), dtype=torch.uint8, device=x.device)
        st = L.ts_w2v_layernorm_bwd_set(x.data_ptr(), res.data_ptr() if res is not None else None, gamma.data_ptr(), dy.data_ptr(), ctx.eps, rows, c,
                                        dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), _s(x))        # dgamma / dbeta written: no fill launches, one reduce
        _lib.check(st, "ts_w2v_layernorm_bwd_set")
        return dx, (dx if res is not None else None), dg, db, None


class BiasGelu(torch.autograd.Function):
    """gelu(z + b), erf form; b may be None."""

    @staticmethod
    def forward(ctx, z, b):
        z = _f32c(z)
        c = z.shape[-1]
        y = torch.empty_like(z)
        _lib.check(_lib.lib().ts_w2v_gelu_fwd(z.data_ptr(), b.data_ptr() if b is not None else None, c, y.data_ptr(), z.numel(), _s(z)), "ts_w2v_gelu_fwd")
        ctx.save_for_backward(z, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, b = ctx.saved_tensors
        dy = _f32c(dy)
        c = z.shape[-1]
        dz = torch.empty_like(z)
        _lib.check(_lib.lib().ts_w2v_gelu_bwd(z.data_ptr(), b.data_ptr() if b is not None else None, c, dy.data_ptr(), dz.data_ptr(), z.numel(), _s(z)),
                   "ts_w2v_gelu_bwd")
        db = _colsum(dz, z.numel() // c, c) if (b is not None and ctx.needs_input_grad[1]) else None
        return dz, db


class Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32c(a), _f32c(b)
        y = torch.empty_like(a)
        _lib.check(_lib.lib().ts_w2v_add(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _s(a)), "ts_w2v_add")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class Dropout(torch.autograd.Function):
    """nn.Dropout on the device's Philox stream (ts_train_dropout): the backward re-draws the mask from the seed."""

    @staticmethod
    def forward(ctx, x, p, seed):
        x = _f32c(x)
        c = x.shape[-1]
        y = torch.empty_like(x)
        st = _lib.lib().ts_train_dropout(x.data_ptr(), y.data_ptr(), x.numel() // c, c, c, float(p), int(seed), None, 0, _s(x))
        _lib.check(st, "ts_train_dropout")
        ctx.p, ctx.seed = p, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _f32c(dy)
        c = dy.shape[-1]
        dx = torch.empty_like(dy)
        st = _lib.lib().ts_train_dropout(dy.data_ptr(), dx.data_ptr(), dy.numel() // c, c, c, float(ctx.p), int(ctx.seed), None, 0, _s(dy))
        _lib.check(st, "ts_train_dropout")
        return dx, None, None


def dropout(x: Tensor, p: float) -> Tensor:
    return Dropout.apply(x, p, next_seed()) if p > 0.0 else x


class MaskRows(torch.autograd.Function):
    """hidden_states[~attention_mask] = 0 (rows >= key_len[clip]); linear, so the backward is the same masking of the gradient."""

    @staticmethod
    def forward(ctx, x, key_len):
        y = _f32c(x).clone()
        b, t, c = y.shape
        _lib.check(_lib.lib().ts_w2v_mask_rows(y.data_ptr(), b, t, c, key_len.data_ptr(), _s(y)), "ts_w2v_mask_rows")
        ctx.save_for_backward(key_len)
        return y

    @staticmethod
    def backward(ctx, dy):
        (key_len,) = ctx.saved_tensors
        dx = _f32c(dy).clone()
        b, t, c = dx.shape
        _lib.check(_lib.lib().ts_w2v_mask_rows(dx.data_ptr(), b, t, c, key_len.data_ptr(), _s(dx)), "ts_w2v_mask_rows")
        return dx, None


class MaskEmbed(torch.autograd.Function):
    """hidden_states[mask_time_indices] = masked_spec_embed (Wav2Vec2Model._mask_hidden_states)."""

    @staticmethod
    def forward(ctx, x, mask, embed):
        y = _f32c(x).clone()
        c = y.shape[-1]
        _lib.check(_lib.lib().ts_w2v_mask_embed(y.data_ptr(), mask.data_ptr(), _f32c(embed).data_ptr(), None, y.numel() // c, c, _s(y)), "ts_w2v_mask_embed")
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dx = _f32c(dy).clone()
        c = dx.shape[-1]
        de = torch.zeros(c, dtype=torch.float32, device=dx.device)
        _lib.check(_lib.lib().ts_w2v_mask_embed(dx.data_ptr(), mask.data_ptr(), None, de.data_ptr(), dx.numel() // c, c, _s(dx)), "ts_w2v_mask_embed")
        return dx, None, de


class Attention(torch.autograd.Function):
    """Multi-head self-attention on a fused qkv tensor [B, T, 3C] (q | k | v thirds, head h = columns [h hd, (h + 1) hd) of each):
    ctx = dropout(softmax(q k^T / sqrt(hd), keys >= key_len masked)) v.  The probabilities are kept for the backward."""

    @staticmethod
    def forward(ctx, qkv, key_len, heads, p_drop, seed):
        qkv = _f32c(qkv)
        b, t, c3 = qkv.shape
        c = c3 // 3
        hd = c // heads
        L = _lib.lib()
        tp = (t + 3) // 4 * 4                       # row pitch of the score / probability rows: 16-byte rows keep every product on vector loads
        p = torch.empty(b, heads, t, tp, dtype=torch.float32, device=qkv.device)
        # scores[b][h][q][k] = q . k
        _gemm(qkv, c3, 1, qkv, 1, c3, p, tp, t, t, hd, sa=t * c3, sa2=hd, sb=t * c3, sb2=hd, sc=heads * t * tp, sc2=t * tp, batch=b, batch2=heads, b_off=c)
        _lib.check(L.ts_w2v_softmax_fwd(p.data_ptr(), key_len.data_ptr() if key_len is not None else None, b, heads, t, tp, hd ** -0.5, _s(p)),
                   "ts_w2v_softmax_fwd")
        pd = p
        if p_drop > 0.0:
            pd = torch.empty_like(p)                 # its pitch columns are never multiplied into a stored output
            _lib.check(L.ts_train_dropout(p.data_ptr(), pd.data_ptr(), b * heads * t, t, tp, float(p_drop), int(seed), None, 0, _s(p)), "ts_train_dropout")
        out = torch.empty(b, t, c, dtype=torch.float32, device=qkv.device)
        _gemm(pd, tp, 1, qkv, c3, 1, out, c, t, hd, t, sa=heads * t * tp, sa2=t * tp, sb=t * c3, sb2=hd, sc=t * c, sc2=hd, batch=b, batch2=heads, b_off=2 * c)
        ctx.save_for_backward(qkv, p, pd if p_drop > 0.0 else None)
        ctx.geom = (heads, p_drop, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, p, pd = ctx.saved_tensors
        heads, p_drop, seed = ctx.geom
        dout = _f32c(dout)
        b, t, c3 = qkv.shape
        c = c3 // 3
        hd = c // heads
        tp = p.shape[-1]
        L = _lib.lib()
        pv = pd if pd is not None else p
        dqkv = torch.empty_like(qkv)
        bh = dict(batch=b, batch2=heads)
        # dV[k][d] = sum_q P'[q][k] dout[q][d]
        _gemm(pv, 1, tp, dout, c, 1, dqkv, c3, t, hd, t, sa=heads * t * tp, sa2=t * tp, sb=t * c, sb2=hd, sc=t * c3, sc2=hd, c_off=2 * c, **bh)
        # dP'[q][k] = sum_d dout[q][d] v[k][d]
        dp = torch.empty_like(p)
        _gemm(dout, c, 1, qkv, 1, c3, dp, tp, t, t, hd, sa=t * c, sa2=hd, sb=t * c3, sb2=hd, sc=heads * t * tp, sc2=t * tp, b_off=2 * c, **bh)
        if p_drop > 0.0:
            _lib.check(L.ts_train_dropout(dp.data_ptr(), dp.data_ptr(), b * heads * t, t, tp, float(p_drop), int(seed), None, 0, _s(dp)), "ts_train_dropout")
        _lib.check(L.ts_w2v_softmax_bwd(p.data_ptr(), dp.data_ptr(), b * heads * t, t, tp, hd ** -0.5, _s(dp)), "ts_w2v_softmax_bwd")
        # dQ[q][d] = sum_k dS[q][k] k[k][d];  dK[k][d] = sum_q dS[q][k] q[q][d]
        _gemm(dp, tp, 1, qkv, c3, 1, dqkv, c3, t, hd, t, sa=heads * t * tp, sa2=t * tp, sb=t * c3, sb2=hd, sc=t * c3, sc2=hd, b_off=c, **bh)
        _gemm(dp, 1, tp, qkv, c3, 1, dqkv, c3, t, hd, t, sa=heads * t * tp, sa2=t * tp, sb=t * c3, sb2=hd, sc=t * c3, sc2=hd, c_off=c, **bh)
        return dqkv, None, None, None, None


class AttentionFused(torch.autograd.Function):
    """The same attention in mixed precision WITHOUT the [T][T] matrices (csrc/w2v_attn_train.hip; head_dim 64): bf16 q / k / v / dO / probabilities, f32
    softmax arithmetic, accumulation and results; the dropout mask is ts_train_dropout's (same seed -> same mask as `Attention`), re-drawn in the two backward
    kernels; saved for the backward: the bf16 qkv, the f32 output and one f32 per (head, query)."""

    @staticmethod
    def supported(c: int, heads: int) -> bool:
        return c % heads == 0 and c // heads == 64

    @staticmethod
    def forward(ctx, qkv, key_len, heads, p_drop, seed):
        qkv = _f32c(qkv)
        b, t, c3 = qkv.shape
        c = c3 // 3
        q16, _ = _cast(qkv.view(b * t, c3), b * t, c3, True, False)
        out = torch.empty(b, t, c, dtype=torch.float32, device=qkv.device)
        lse2 = torch.empty(b, heads, t, dtype=torch.float32, device=qkv.device)
        L = _lib.lib()
        ws = torch.empty(L.ts_w2v_attention_train_fwd_workspace(b, t, c, heads), dtype=torch.uint8, device=qkv.device) if p_drop > 0 else None
        st = L.ts_w2v_attention_train_fwd(q16.data_ptr(), b, t, c, heads, key_len.data_ptr() if key_len is not None else None, float(p_drop), int(seed),
                                          out.data_ptr(), lse2.data_ptr(), ws.data_ptr() if ws is not None else None, _s(qkv))
        _lib.check(st, "ts_w2v_attention_train_fwd")
        ctx.save_for_backward(q16, out, lse2, key_len, ws)                       # ws: the mask bits (the backward would re-draw them otherwise)
        ctx.geom = (b, t, c, heads, float(p_drop), int(seed))
        return out

    @staticmethod
    def backward(ctx, dout):
        q16, out, lse2, key_len, mask = ctx.saved_tensors
        b, t, c, heads, p_drop, seed = ctx.geom
        dout = _f32c(dout)
        L = _lib.lib()
        ws = torch.empty(L.ts_w2v_attention_train_bwd_workspace(b, t, c, heads), dtype=torch.uint8, device=dout.device)
        dqkv = torch.empty(b, t, 3 * c, dtype=torch.float32, device=dout.device)
        st = L.ts_w2v_attention_train_bwd(q16.data_ptr(), b, t, c, heads, key_len.data_ptr() if key_len is not None else None, p_drop, seed, dout.data_ptr(),
                                          out.data_ptr(), lse2.data_ptr(), mask.data_ptr() if mask is not None else None, dqkv.data_ptr(), ws.data_ptr(), _s(dout))
        _lib.check(st, "ts_w2v_attention_train_bwd")
        return dqkv, None, None, None, None


MIXED_POSCONV = True        # mixed mode: the positional conv's forward and data gradient on the bf16 matrix-core kernel (False: f32 products, for A/B)
FUSED_ATTENTION = True      # mixed mode: AttentionFused where it applies (False: the materialised-probabilities path, for A/B)


def attention(qkv: Tensor, key_len, heads: int, p_drop: float, seed: int) -> Tensor:
    """Self-attention of the training path in the current precision mode."""
    if _MIXED and FUSED_ATTENTION and AttentionFused.supported(qkv.shape[-1] // 3, heads):
        return AttentionFused.apply(qkv, key_len, heads, p_drop, seed)
    return Attention.apply(qkv, key_len, heads, p_drop, seed)


class PosConvGelu(torch.autograd.Function):
    """y = x + gelu(grouped_conv1d(x, w, padding = k // 2)[..., :T] + b) (Wav2Vec2PositionalConvEmbedding + the residual add).
    wk: the effective (weight-normalised) conv weight as [k][groups][out][in] f32 -- computed from (g, v) by torch ops on the PARAMETERS, so that
    autograd carries d wk back through the weight norm; x [B, T, C] time-major.  The conv and both of its gradients are ONE GEMM launch each: the
    tap loop is the GEMM's outer contraction loop (forward, data gradient) or its second batch level (weight gradient) over a zero-padded copy
    of the rows."""

    @staticmethod
    def _mixed_ok(k: int, cg: int) -> bool:
        return _MIXED and MIXED_POSCONV and cg == 64 and (128 + k - 1) * 144 <= 64 * 1024

    @staticmethod
    def forward(ctx, x, wk, bias):
        x, wk, bias = _f32c(x), _f32c(wk), _f32c(bias)
        b, t, c = x.shape
        k, g, cg, _ = wk.shape
        L = _lib.lib()
        tp = t + k
        ctx.mixed = PosConvGelu._mixed_ok(k, cg)
        if ctx.mixed:
            # mixed precision: the conv on the inference path's matrix-core kernel (bf16 operands, f32 accumulation; csrc/w2v_enc.hip w2v_posconv_mfma_kernel),
            # which also leaves the pre-activation z for the backward; bias + GELU + residual in its epilogue
            w16 = wk.to(torch.bfloat16)
            ws = torch.empty(L.ts_w2v_posconv_train_workspace(b, t, c, k), dtype=torch.uint8, device=x.device)
            y, z = torch.empty_like(x), torch.empty_like(x)
            _lib.check(L.ts_w2v_posconv_train(x.data_ptr(), x.data_ptr(), b, t, c, w16.data_ptr(), bias.data_ptr(), k, g, 0, y.data_ptr(), z.data_ptr(),
                                              ws.data_ptr(), _s(x)), "ts_w2v_posconv_train")
            ctx.save_for_backward(x, z, wk, bias)
            ctx.geom = (b, t, c)
            return y
        xp = torch.empty(b, tp, c, dtype=torch.float32, device=x.device)
        _lib.check(L.ts_w2v_pad_rows(x.data_ptr(), xp.data_ptr(), b, t, tp, k // 2, c, 0, _s(x)), "ts_w2v_pad_rows")
        m = b * tp - k                                                   # rows of the padded row space that have all k taps
        zp = torch.zeros(b, tp, c, dtype=torch.float32, device=x.device)
        # zp[r][g cg + o] = sum_j sum_i xp[r + j][g cg + i] wk[j][g][o][i]
        _gemm(xp, c, 1, wk, 1, cg, zp, c, m, cg, cg, sa=cg, ska=c, sb=cg * cg, skb=g * cg * cg, sc=cg, nkb=k, batch=g)
        z = torch.empty_like(x)
        _lib.check(L.ts_w2v_pad_rows(zp.data_ptr(), z.data_ptr(), b, t, tp, 0, c, 1, _s(x)), "ts_w2v_pad_rows")
        a = torch.empty_like(x)
        _lib.check(L.ts_w2v_gelu_fwd(z.data_ptr(), bias.data_ptr(), c, a.data_ptr(), z.numel(), _s(x)), "ts_w2v_gelu_fwd")
        y = torch.empty_like(x)
        _lib.check(L.ts_w2v_add(x.data_ptr(), a.data_ptr(), y.data_ptr(), x.numel(), _s(x)), "ts_w2v_add")
        ctx.save_for_backward(xp, z, wk, bias)
        ctx.geom = (b, t, c)
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, z, wk, bias = ctx.saved_tensors
        b, t, c = ctx.geom
        k, g, cg, _ = wk.shape
        dy = _f32c(dy)
        L = _lib.lib()
        tp = t + k
        m = b * tp - k
        x = xp if ctx.mixed else None                    # (mixed mode: the node kept x, not its padded copy)
        dz = torch.empty_like(dy)
        _lib.check(L.ts_w2v_gelu_bwd(z.data_ptr(), bias.data_ptr(), c, dy.data_ptr(), dz.data_ptr(), dz.numel(), _s(dy)), "ts_w2v_gelu_bwd")
        db = _colsum(dz, b * t, c) if ctx.needs_input_grad[2] else None
        if not ctx.mixed:
            # d zp in the padded row space behind k - 1 zero rows: dbuf[k - 1 + r] = d zp[r]
            dbuf = torch.empty((k - 1) + b * tp, c, dtype=torch.float32, device=dy.device)
            dbuf[: k - 1].zero_()
            dzp = dbuf[k - 1:].view(b, tp, c)
            _lib.check(L.ts_w2v_pad_rows(dz.data_ptr(), dzp.data_ptr(), b, t, tp, 0, c, 0, _s(dy)), "ts_w2v_pad_rows")
        dx = dwk = None
        if ctx.needs_input_grad[0] and ctx.mixed:
            # dx = dy + conv^T(dz): the same matrix-core kernel over dz with the taps flipped and each tap's [out][in] block transposed
            wb16 = wk.flip(0).transpose(2, 3).contiguous().to(torch.bfloat16)
            ws = torch.empty(L.ts_w2v_posconv_train_workspace(b, t, c, k), dtype=torch.uint8, device=dy.device)
            dx = torch.empty_like(dy)
            _lib.check(L.ts_w2v_posconv_train(dz.data_ptr(), dy.data_ptr(), b, t, c, wb16.data_ptr(), None, k, g, 1, dx.data_ptr(), None, ws.data_ptr(), _s(dy)),
                       "ts_w2v_posconv_train")
        elif ctx.needs_input_grad[0]:
            # d xp[q][g cg + i] = sum_j' sum_o dbuf[q + j'][g cg + o] wk[k - 1 - j'][g][o][i]
            dxp = torch.empty(b, tp, c, dtype=torch.float32, device=dy.device)
            _gemm(dbuf, c, 1, wk, cg, 1, dxp, c, b * tp, cg, cg, sa=cg, ska=c, sb=cg * cg, skb=-g * cg * cg, sc=cg, nkb=k, batch=g,
                  b_off=(k - 1) * g * cg * cg)
            dconv = torch.empty_like(dy)
            _lib.check(L.ts_w2v_pad_rows(dxp.data_ptr(), dconv.data_ptr(), b, t, tp, k // 2, c, 1, _s(dy)), "ts_w2v_pad_rows")
            dx = torch.empty_like(dy)
            _lib.check(L.ts_w2v_add(dy.data_ptr(), dconv.data_ptr(), dx.data_ptr(), dy.numel(), _s(dy)), "ts_w2v_add")
        if ctx.needs_input_grad[1] and ctx.mixed:
            # the same sums on the matrix cores (bf16 operands): one workgroup per (8 taps, group), no partials
            dwk = torch.empty_like(wk)
            ws = torch.empty(L.ts_w2v_posconv_wgrad_workspace(b, t, c, k), dtype=torch.uint8, device=dy.device)
            _lib.check(L.ts_w2v_posconv_wgrad(dz.data_ptr(), x.data_ptr(), b, t, c, k, g, dwk.data_ptr(), ws.data_ptr(), _s(dy)), "ts_w2v_posconv_wgrad")
        elif ctx.needs_input_grad[1]:
            # d wk[j][g][o][i] = sum_r d zp[r][g cg + o] xp[r + j][g cg + i]   (rows between the clips hold d zp = 0)
            dwk = torch.empty_like(wk)
            _gemm(dzp, 1, c, xp, c, 1, dwk, cg, cg, cg, m, sa=cg, sa2=0, sb=cg, sb2=c, sc=cg * cg, sc2=g * cg * cg, batch=g, batch2=k)
        return dx, dwk, db


def compute_mask_indices(batch: int, t: int, mask_prob: float, mask_length: int, lengths=None, min_masks: int = 0) -> np.ndarray:
    """SpecAugment-style time masks of transformers' `_compute_mask_indices` (modeling_wav2vec2.py; the published algorithm, numpy's global
    RNG like the original): per clip, `int(mask_prob * len / mask_length + eps)` spans (at least `min_masks`) of `mask_length` frames at distinct
    random starts; clips with fewer spans repeat their first one.  Returns bool [batch, t]."""
    if mask_length < 1:
        raise ValueError("`mask_length` has to be bigger than 0.")
    if mask_length > t:
        raise ValueError(f"`mask_length` has to be smaller than `sequence_length`, but got `mask_length`: {mask_length} and `sequence_length`: {t}`")
    eps = np.random.rand(1).item()

    def n_spans(n):
        k = max(int(mask_prob * n / mask_length + eps), min_masks)
        if k * mask_length > t:
            k = t // mask_length
        if n - (mask_length - 1) < k:
            k = max(n - (mask_length - 1), 0)
        return k

    lens = [t] * batch if lengths is None else [int(v) for v in lengths]
    mask = np.zeros((batch, t), dtype=bool)
    k_max = n_spans(t)
    if k_max == 0:
        return mask
    for i, n in enumerate(lens):
        k = n_spans(n)
        starts = np.random.choice(np.arange(n - (mask_length - 1)), k, replace=False)
        fill = t - 1 if len(starts) == 0 else starts[0]
        starts = np.concatenate([starts, np.full(k_max - k, fill, dtype=np.int64)]).astype(np.int64)
        idx = (starts[:, None] + np.arange(mask_length)[None, :]).reshape(-1)
        mask[i, np.minimum(idx, t - 1)] = True
    return mask


def train_forward(adapt, audio: Tensor, lengths: Optional[Tensor]) -> Tensor:
    """Training-mode Wav2Vec2Model.forward -> last_hidden_state [B, T', C] (f32, time-major).  `adapt`: the HuggingFaceEncoderAdapt module
    (its `original_encoder` owns the parameters).  `adapt.train_precision`: "fp32" (default: the reference's arithmetic, every product on the f32
    matrix-core GEMM) or "bf16" (mixed precision: the linear layers -- 92 % of the step's FLOPs -- multiply bf16 operands with f32 accumulation,
    LinearMixed; attention, positional conv, LayerNorm, GELU, dropout, master weights and all gradients stay f32)."""
    global _MIXED
    _MIXED = getattr(adapt, "train_precision", "fp32") == "bf16"
    try:
        return _train_forward(adapt, audio, lengths)
    finally:
        _MIXED = False


def _train_forward(adapt, audio: Tensor, lengths: Optional[Tensor]) -> Tensor:
    from .encoder import feat_extract_output_lengths
    enc = adapt.original_encoder
    cfg = enc.config
    dev = audio.device
    if getattr(cfg, "model_type", "wav2vec2") != "wav2vec2":
        # hubert / data2vec-audio run the inference path (huggingface/encoder.py); their training-mode forward (no feature-projection LayerNorm,
        # stacked positional convs) has no autograd nodes here
        raise NotImplementedError(f"HIP fine-tuning path: model_type={cfg.model_type!r} is inference-only (wav2vec2 checkpoints fine-tune)")
    if getattr(cfg, "add_adapter", False):
        raise NotImplementedError("HIP fine-tuning path: add_adapter=True is inference-only (the adapter layers have no backward here)")
    if float(getattr(cfg, "mask_feature_prob", 0.0)) > 0.0 and getattr(cfg, "apply_spec_augment", True):
        raise NotImplementedError("wav2vec2 HIP training path: mask_feature_prob > 0 is not supported")
    unfrozen = [n for n, p in enc.feature_extractor.named_parameters() if p.requires_grad]
    if unfrozen:
        # the reference freezes the conv feature extractor in __init__ (compatibility.py:27-28) and never trains it; autograd WOULD train it
        # if a user flipped requires_grad afterwards -- this path has no backward for those convolutions, so say so instead of silently
        # returning no gradient
        raise NotImplementedError("wav2vec2 HIP training path: the conv feature extractor is frozen (freeze_feature_encoder); "
                                  f"{len(unfrozen)} of its parameters have requires_grad=True (first: feature_extractor.{unfrozen[0]})")
    plan = adapt._plan_frozen(dev)
    with torch.no_grad():                                   # frozen conv feature extractor (compatibility.py:27-28): the inference kernels
        feats = plan.feature_extractor(audio)
    b, t, _ = feats.shape
    c, heads = int(cfg.hidden_size), int(cfg.num_attention_heads)
    eps = float(cfg.layer_norm_eps)
    fp, en = enc.feature_projection, enc.encoder
    h = LayerNorm.apply(feats, None, fp.layer_norm.weight, fp.layer_norm.bias, eps)
    h = linear(h, fp.projection.weight, fp.projection.bias)
    h = dropout(h, float(cfg.feat_proj_dropout))
    key_len = None
    if lengths is not None:
        key_len = feat_extract_output_lengths(cfg.conv_kernel, cfg.conv_stride, lengths.to(dev).long()).to(torch.int32).contiguous()
    if getattr(cfg, "apply_spec_augment", True) and float(cfg.mask_time_prob) > 0.0:
        m = compute_mask_indices(b, t, float(cfg.mask_time_prob), int(cfg.mask_time_length), None if key_len is None else key_len.tolist(),
                                 int(getattr(cfg, "mask_time_min_masks", 2)))
        if m.any():
            h = MaskEmbed.apply(h, torch.from_numpy(m.astype(np.uint8)).to(dev), enc.masked_spec_embed)
    if key_len is not None:
        h = MaskRows.apply(h, key_len)
    # positional conv: weight norm over dim 2 on the parameters (torch ops on [C][C/g][k] only), conv + GELU + residual on the kernels
    conv = en.pos_conv_embed.conv
    if hasattr(conv, "parametrizations"):
        wg, wv = conv.parametrizations.weight.original0, conv.parametrizations.weight.original1
    else:
        wg, wv = conv.weight_g, conv.weight_v
    g = int(cfg.num_conv_pos_embedding_groups)
    kpos = int(cfg.num_conv_pos_embeddings)
    w_eff = wg * wv / wv.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    wk = w_eff.view(g, c // g, c // g, kpos).permute(3, 0, 1, 2).contiguous()
    h = PosConvGelu.apply(h, wk, conv.bias)
    stable = bool(getattr(cfg, "do_stable_layer_norm", False))
    p_hid, p_act, p_att = float(cfg.hidden_dropout), float(cfg.activation_dropout), float(cfg.attention_dropout)
    if not stable:
        h = LayerNorm.apply(h, None, en.layer_norm.weight, en.layer_norm.bias, eps)
    h = dropout(h, p_hid)
    for layer in en.layers:
        if torch.rand([]).item() < float(cfg.layerdrop):            # LayerDrop: transformers draws torch.rand([]) per layer in training
            continue
        att, ff = layer.attention, layer.feed_forward
        wqkv = torch.cat([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight], 0)
        bqkv = torch.cat([att.q_proj.bias, att.k_proj.bias, att.v_proj.bias], 0)

        def attend(x):
            ctxt = attention(linear(x, wqkv, bqkv), key_len, heads, p_att, next_seed() if p_att > 0 else 0)
            return dropout(linear(ctxt, att.out_proj.weight, att.out_proj.bias), p_hid)

        def ffn(x):
            z = linear(x, ff.intermediate_dense.weight, None)
            w2 = ff.output_dense.weight
            if _MIXED and FUSED_FFN_ACT and FFNActLinear.supported(w2.shape[1], w2.shape[0]):
                return dropout(FFNActLinear.apply(z, ff.intermediate_dense.bias, w2, ff.output_dense.bias, p_act, next_seed() if p_act > 0 else 0), p_hid)
            a = dropout(BiasGelu.apply(z, ff.intermediate_dense.bias), p_act)
            return dropout(linear(a, w2, ff.output_dense.bias), p_hid)

        if stable:                                                  # pre-LN: h += attn(LN(h)); h += ffn(LN(h))
            h = Add.apply(h, attend(LayerNorm.apply(h, None, layer.layer_norm.weight, layer.layer_norm.bias, eps)))
            h = Add.apply(h, ffn(LayerNorm.apply(h, None, layer.final_layer_norm.weight, layer.final_layer_norm.bias, eps)))
        else:                                                       # post-LN: h = LN(h + attn(h)); h = LN(h + ffn(h))
            h = LayerNorm.apply(attend(h), h, layer.layer_norm.weight, layer.layer_norm.bias, eps)
            h = LayerNorm.apply(ffn(h), h, layer.final_layer_norm.weight, layer.final_layer_norm.bias, eps)
    if stable:
        h = LayerNorm.apply(h, None, en.layer_norm.weight, en.layer_norm.bias, eps)
    return h
