"""wav2vec2 encoder on the HIP kernels of csrc/w2v_enc.hip.

The reference hands the whole encoder to transformers (`_HuggingFaceEncoderAdapt.forward`,
src/thunder/huggingface/compatibility.py:31-42: `self.original_encoder(audio, attention_mask=...)`, then
`last_hidden_state.transpose(-1, -2)` and `_get_feat_extract_output_lengths`).  Here the transformers module only OWNS the
weights (same attribute name `original_encoder`, same state-dict keys, so checkpoints and fine-tuning code see no
difference); the arithmetic runs in `Wav2Vec2Plan`, which keeps packed fp32 device copies of the weights and issues one C-ABI
call per stage.  Both published families (group-norm / post-LN: wav2vec2-base-960h, -large-960h; layer-norm / pre-LN: -large-lv60, xlsr);
training mode (fine-tuning with the conv feature extractor frozen) runs through huggingface/train.py; no CPU fallback."""
from __future__ import annotations

import os

import ctypes as C
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from .. import _lib, tensors as _t
from ..blocks import _PackedCache

__all__ = ["Wav2Vec2Plan", "HuggingFaceEncoderAdapt", "feat_extract_output_lengths"]


def feat_extract_output_lengths(conv_kernel, conv_stride, lengths: torch.Tensor, adapter_layers: int = 0, adapter_stride: int = 2) -> torch.Tensor:
    """transformers' `_get_feat_extract_output_lengths`: floor((len - k) / s) + 1 per conv layer (integer tensor in, out); with an adapter behind the
    encoder (config.add_adapter) also floor((len - 1) / adapter_stride) + 1 per adapter layer."""
    out = lengths
    for k, s in zip(conv_kernel, conv_stride):
        out = torch.div(out - k, s, rounding_mode="floor") + 1
    for _ in range(adapter_layers):
        out = torch.div(out - 1, adapter_stride, rounding_mode="floor") + 1
    return out


# AutoModelForCTC families with a HIP path (the reference loads any of them, huggingface/compatibility.py:77): wav2vec2 (both published
# families), hubert (the same layers; the feature projection's LayerNorm is optional) and data2vec-audio (the reference's own test,
# tests/huggingface/test_module_huggingface.py:107-110: layer-norm conv feature extractor, post-LN encoder, the positional embedding as a
# stack of grouped convs each followed by an affine-free LayerNorm and GELU); unispeech / unispeech-sat (UniSpeechModel / UniSpeechSatModel run the wav2vec2
# encoder arithmetic unchanged: tests/test_oracle_w2v.py checks the oracle against both).  Others (wavlm's gated relative-position attention, sew's
# squeezed encoder, wav2vec2-conformer ...) have layers this library holds no kernels for and raise.
SUPPORTED_MODEL_TYPES = ("wav2vec2", "hubert", "data2vec-audio", "unispeech", "unispeech-sat")


def _check_config(cfg) -> None:
    bad = []
    if getattr(cfg, "model_type", "wav2vec2") not in SUPPORTED_MODEL_TYPES:
        bad.append(f"model_type={cfg.model_type!r} (supported: {', '.join(SUPPORTED_MODEL_TYPES)})")
    if getattr(cfg, "conv_pos_batch_norm", False):
        bad.append("conv_pos_batch_norm=True")
    if getattr(cfg, "feat_extract_norm", "group") not in ("group", "layer"):
        bad.append(f"feat_extract_norm={cfg.feat_extract_norm!r}")
    if getattr(cfg, "hidden_act", "gelu") != "gelu" or getattr(cfg, "feat_extract_activation", "gelu") != "gelu":
        bad.append("activation != gelu")
    if getattr(cfg, "position_embeddings_type", None) not in (None, "absolute") and hasattr(cfg, "position_embeddings_type"):
        bad.append(f"position_embeddings_type={cfg.position_embeddings_type!r}")
    if bad:
        raise NotImplementedError("wav2vec2 HIP path: unsupported configuration: " + ", ".join(bad))


class Wav2Vec2Plan:
    """Packed weights + launch sequence for one set of encoder weights (HF state-dict keys, see oracle/w2v.py).
    precision "fp32": every GEMM in fp32; "bf16": GEMM operands in bf16 (fp32 accumulation, fp32 residual stream /
    normalisations / softmax) -- each launch also writes the bf16 copy its consumer multiplies with."""

    def __init__(self, cfg, sd: Dict[str, torch.Tensor], device, precision: str = "bf16", feature_extractor_only: bool = False):
        """`feature_extractor_only`: pack the conv feature extractor's weights and nothing else -- the plan of the training path's frozen
        front (huggingface/train.py); `forward` is then unavailable, `feature_extractor` works."""
        _check_config(cfg)
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"precision must be 'fp32' or 'bf16', got {precision!r}")
        self.prec = 1 if precision == "bf16" else 0
        f = lambda k: sd[k].detach().to(device=device, dtype=torch.float32).contiguous()
        gw = (lambda t: t.to(torch.bfloat16).contiguous()) if self.prec else (lambda t: t.contiguous())     # GEMM operand
        self.device = torch.device(device)
        self.kernels = [int(k) for k in cfg.conv_kernel]
        self.strides = [int(s) for s in cfg.conv_stride]
        self.dims = [int(c) for c in cfg.conv_dim]
        self.hidden = int(cfg.hidden_size)
        self.heads = int(cfg.num_attention_heads)
        self.n_layers = int(cfg.num_hidden_layers)
        self.kpos = int(cfg.num_conv_pos_embeddings)
        self.groups = int(cfg.num_conv_pos_embedding_groups)
        self.eps = float(cfg.layer_norm_eps)
        self.model_type = getattr(cfg, "model_type", "wav2vec2")
        self.d2v = self.model_type == "data2vec-audio"
        # lv60 / xlsr family; Data2VecAudioConvLayer is always conv -> LayerNorm -> GELU and its config carries no feat_extract_norm
        self.layer_norm_convs = self.d2v or getattr(cfg, "feat_extract_norm", "group") == "layer"
        self.stable_ln = bool(getattr(cfg, "do_stable_layer_norm", False))
        self.fp_has_ln = bool(getattr(cfg, "feat_proj_layer_norm", True))                  # HubertFeatureProjection
        opt = lambda k: f(k) if k in sd else None
        self.conv_b = [opt(f"feature_extractor.conv_layers.{i}.conv.bias") if getattr(cfg, "conv_bias", False) else None
                       for i in range(len(self.kernels))]
        self.conv_ln = [(f(f"feature_extractor.conv_layers.{i}.layer_norm.weight"), f(f"feature_extractor.conv_layers.{i}.layer_norm.bias"))
                        for i in range(len(self.kernels))] if self.layer_norm_convs else None
        self.w0 = f("feature_extractor.conv_layers.0.conv.weight").reshape(self.dims[0], self.kernels[0]).contiguous()
        self.gn_w = f("feature_extractor.conv_layers.0.layer_norm.weight")       # GroupNorm affine ("group" family)
        self.gn_b = f("feature_extractor.conv_layers.0.layer_norm.bias")
        # [c_out][c_in][k] -> [c_out][k][c_in]: consecutive taps are consecutive K columns of one GEMM
        self.conv_w = [gw(f(f"feature_extractor.conv_layers.{i}.conv.weight").permute(0, 2, 1))
                       for i in range(1, len(self.kernels))]
        self.feature_extractor_only = bool(feature_extractor_only)
        self.layers = []
        if self.feature_extractor_only:
            return
        self.fp_ln = (f("feature_projection.layer_norm.weight"), f("feature_projection.layer_norm.bias")) if self.fp_has_ln else None
        self.fp_w, self.fp_b = gw(f("feature_projection.projection.weight")), f("feature_projection.projection.bias")
        cg = self.hidden // self.groups
        if self.d2v:
            # Data2VecAudioPositionalConvEmbedding: num_conv_pos_embeddings LAYERS of conv_pos_kernel_size taps, plain weights
            self.pos_layers = int(cfg.num_conv_pos_embeddings)
            self.kpos = int(cfg.conv_pos_kernel_size)
            self.pos_stack = []
            for i in range(self.pos_layers):
                q = f"encoder.pos_conv_embed.layers.{i}.conv."
                self.pos_stack.append((gw(f(q + "weight").view(self.groups, cg, cg, self.kpos).permute(3, 0, 1, 2)), f(q + "bias")))
            self.unit_ln = (torch.ones(self.hidden, device=self.device), torch.zeros(self.hidden, device=self.device))   # elementwise_affine=False
        else:
            # weight_norm(dim=2): w[:, :, j] = g[j] v[:, :, j] / ||v[:, :, j]||
            p = "encoder.pos_conv_embed.conv."
            if p + "parametrizations.weight.original0" in sd:
                g, v = f(p + "parametrizations.weight.original0"), f(p + "parametrizations.weight.original1")
            else:
                g, v = f(p + "weight_g"), f(p + "weight_v")
            w_eff = g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()                    # [C][C/g][k]
            self.pos_w = gw(w_eff.view(self.groups, cg, cg, self.kpos).permute(3, 0, 1, 2))   # [k][g][out][in]
            self.pos_b = f(p + "bias")
        self.enc_ln = (f("encoder.layer_norm.weight"), f("encoder.layer_norm.bias"))
        # Wav2Vec2Adapter behind the encoder (config.add_adapter): optional projection + LayerNorm, then strided conv (padding 1, 2x channels) + GLU layers.
        # Its projection and convolutions run on the f32 GEMM in both precision modes (three short layers at an eighth of the frame rate and below).
        self.adapter = bool(getattr(cfg, "add_adapter", False))
        if self.adapter:
            self.ad_k, self.ad_s = int(cfg.adapter_kernel_size), int(cfg.adapter_stride)
            self.ad_proj = None
            if "adapter.proj.weight" in sd:
                self.ad_proj = (f("adapter.proj.weight"), f("adapter.proj.bias"), (f("adapter.proj_layer_norm.weight"), f("adapter.proj_layer_norm.bias")))
            self.ad_layers = [(f(f"adapter.layers.{i}.conv.weight").permute(0, 2, 1).contiguous(), f(f"adapter.layers.{i}.conv.bias"))
                              for i in range(int(cfg.num_adapter_layers))]
        for i in range(self.n_layers):
            q = f"encoder.layers.{i}."
            self.layers.append(dict(
                wqkv=gw(torch.cat([f(q + f"attention.{n}_proj.weight") for n in "qkv"], 0)),
                bqkv=torch.cat([f(q + f"attention.{n}_proj.bias") for n in "qkv"], 0).contiguous(),
                wo=gw(f(q + "attention.out_proj.weight")), bo=f(q + "attention.out_proj.bias"),
                ln1=(f(q + "layer_norm.weight"), f(q + "layer_norm.bias")),
                w1=gw(f(q + "feed_forward.intermediate_dense.weight")), b1=f(q + "feed_forward.intermediate_dense.bias"),
                w2=gw(f(q + "feed_forward.output_dense.weight")), b2=f(q + "feed_forward.output_dense.bias"),
                ln2=(f(q + "final_layer_norm.weight"), f(q + "final_layer_norm.bias"))))

    def _frag(self, w: torch.Tensor):
        """The bf16 GEMM weight `w` [n][k] in MFMA B-fragment order (ts_gemm_nt_pack_w), packed once per weight: the GEMM kernel then
        loads its B operand from L2 straight into registers.  None in fp32 mode or for shapes the packed kernel does not take."""
        if not self.prec:
            return None
        cache = self.__dict__.setdefault("_frags", {})
        key = w.data_ptr()
        if key not in cache:
            w2 = w.reshape(w.shape[0], -1)
            n, k = w2.shape
            out = None
            if n % 16 == 0 and k % 32 == 0 and w2.is_contiguous():
                out = torch.empty_like(w2)
                st = _lib.lib().ts_gemm_nt_pack_w(w2.data_ptr(), k, n, k, out.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream)
                if st == _lib.TS_EUNSUPPORTED:
                    out = None
                else:
                    _lib.check(st, "ts_gemm_nt_pack_w")
            cache[key] = (out, w)            # holds w so that the key stays unique
        return cache[key][0]

    # ---- launch helpers: every helper returns (fp32 result, GEMM operand for the next stage) ------------------------
    def _buf(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def _op(self, *shape):
        """Buffer for the GEMM-operand copy of a result (None in fp32 mode: the fp32 result is the operand)."""
        return self._buf(*shape, dtype=torch.bfloat16) if self.prec else None

    @staticmethod
    def _ptr(t):
        return t.data_ptr() if t is not None else None

    def _linear(self, L, stream, x_op, w, bias, act=0, want_op=False, res=None, into=None):
        """want_op: the consumer is another GEMM -- in bf16 mode only the bf16 copy is kept (the fp32 buffer is scratch).
        into: fp32 tensor the product is accumulated into in place (the residual stream; beta = 1 inside the GEMM)."""
        b, t, k = x_op.shape
        n = w.shape[0]
        y = into if into is not None else self._buf(b, t, n)
        res = into if into is not None else res
        y_op = self._op(b, t, n) if want_op else None
        st = L.ts_w2v_linear_fwd(x_op.data_ptr(), k, w.data_ptr(), self._ptr(bias), self._ptr(res), n, y.data_ptr(), n, self._ptr(y_op),
                                 b * t, n, k, act | (2 if y_op is not None else 0), self.prec, self._ptr(self._frag(w)), stream)
        _lib.check(st, "ts_w2v_linear_fwd")
        return y, (y_op if self.prec else y)

    def _ln(self, L, stream, x, wb, res=None, xbias=None, want_op=True, act=0, eps=None, want_f32=True):
        """want_f32=False (bf16 mode only): the f32 result has no reader -- only the GEMM-operand copy is written."""
        y_op = self._op(*x.shape) if want_op else None
        y = torch.empty_like(x) if (want_f32 or y_op is None) else None
        st = L.ts_w2v_layernorm_fwd(x.data_ptr(), self._ptr(res), self._ptr(xbias), wb[0].data_ptr(), wb[1].data_ptr(),
                                    self.eps if eps is None else eps, x.shape[0] * x.shape[1], x.shape[2], act, self._ptr(y),
                                    self._ptr(y_op), stream)
        _lib.check(st, "ts_w2v_layernorm_fwd")
        return y, (y_op if self.prec else y)

    def feature_extractor(self, audio: torch.Tensor) -> torch.Tensor:
        """[B, n] fp32 -> [B, T', C_last] fp32 (time-major)."""
        return self._feature_extractor(audio)[0]

    def _feature_extractor(self, audio: torch.Tensor, want_op: bool = False):
        """-> (features fp32, their GEMM-operand copy when `want_op`: a feature projection without LayerNorm multiplies them as they are)."""
        L = _lib.lib()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        b, n = audio.shape
        k0, s0, c0 = self.kernels[0], self.strides[0], self.dims[0]
        if n < k0:
            raise RuntimeError(f"wav2vec2: input of {n} samples is shorter than the first conv kernel ({k0})")
        t = (n - k0) // s0 + 1
        ws = self._buf(L.ts_w2v_conv0_workspace_bytes(b, n, c0, k0, s0), dtype=torch.uint8)
        n_conv = len(self.kernels)
        if self.layer_norm_convs:
            # every layer: conv (+ bias) -> LayerNorm over the channels -> GELU; the conv output stays fp32 for the LayerNorm
            h = self._buf(b, t, c0)
            _lib.check(L.ts_w2v_conv0_fwd(audio.data_ptr(), b, n, self.w0.data_ptr(), None, self._ptr(self.conv_b[0]), c0, k0, s0, 1e-5,
                                          h.data_ptr(), None, ws.data_ptr(), stream), "ts_w2v_conv0_fwd")
            h, x_op = self._ln(L, stream, h, self.conv_ln[0], act=1, eps=1e-5, want_op=n_conv > 1 or want_op, want_f32=n_conv == 1)
            for i, w in enumerate(self.conv_w, start=1):
                k, s = self.kernels[i], self.strides[i]
                t_out = (t - k) // s + 1
                if t_out < 1:
                    raise RuntimeError("wav2vec2: input too short for the conv feature extractor")
                y = self._buf(b, t_out, self.dims[i])
                _lib.check(L.ts_w2v_conv_fwd(x_op.data_ptr(), b, t, self.dims[i - 1], w.data_ptr(), self._ptr(self.conv_b[i]), self.dims[i],
                                             k, s, 0, self.prec, y.data_ptr(), None, self._ptr(self._frag(w)), stream), "ts_w2v_conv_fwd")
                h, x_op = self._ln(L, stream, y, self.conv_ln[i], act=1, eps=1e-5, want_op=i < n_conv - 1 or want_op, want_f32=i == n_conv - 1)
                t = t_out
            return h, x_op
        last = n_conv == 1
        h = self._buf(b, t, c0) if (not self.prec or last) else None       # bf16 mode: the next conv only reads the bf16 copy
        h_op = self._op(b, t, c0)
        _lib.check(L.ts_w2v_conv0_fwd(audio.data_ptr(), b, n, self.w0.data_ptr(), self.gn_w.data_ptr(), self.gn_b.data_ptr(), c0, k0,
                                      s0, 1e-5, self._ptr(h), self._ptr(h_op), ws.data_ptr(), stream), "ts_w2v_conv0_fwd")
        x_op = h_op if self.prec else h
        for i, w in enumerate(self.conv_w, start=1):
            k, s = self.kernels[i], self.strides[i]
            t_out = (t - k) // s + 1
            if t_out < 1:
                raise RuntimeError("wav2vec2: input too short for the conv feature extractor")
            last = i == n_conv - 1
            y = self._buf(b, t_out, self.dims[i])
            y_op = None if (last and not want_op) else self._op(b, t_out, self.dims[i])
            _lib.check(L.ts_w2v_conv_fwd(x_op.data_ptr(), b, t, self.dims[i - 1], w.data_ptr(), self._ptr(self.conv_b[i]), self.dims[i], k, s,
                                         1, self.prec, y.data_ptr(), self._ptr(y_op), self._ptr(self._frag(w)), stream), "ts_w2v_conv_fwd")
            h, t = y, t_out
            x_op = y_op if self.prec else y
        return h, x_op

    def forward(self, audio: torch.Tensor, lengths: Optional[torch.Tensor]) -> torch.Tensor:
        """audio [B, n] fp32 on the GPU; lengths = samples per clip when the model was trained with an attention mask
        (`mask_input`), else None.  Returns last_hidden_state [B, T', C] fp32 (time-major)."""
        if self.feature_extractor_only:
            raise RuntimeError("this Wav2Vec2Plan holds the feature extractor's weights only (feature_extractor_only=True)")
        L = _lib.lib()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        feats, ln_op = self._feature_extractor(audio, want_op=not self.fp_has_ln)
        b, t, _ = feats.shape
        c = self.hidden
        if self.fp_has_ln:
            _, ln_op = self._ln(L, stream, feats, self.fp_ln, want_f32=False)
        h, _ = self._linear(L, stream, ln_op, self.fp_w, self.fp_b)
        key_len = None
        if lengths is not None:
            key_len = feat_extract_output_lengths(self.kernels, self.strides, lengths.to(self.device).long()).to(torch.int32).contiguous()
            _lib.check(L.ts_w2v_mask_rows(h.data_ptr(), b, t, c, key_len.data_ptr(), stream), "ts_w2v_mask_rows")
        ws = self._buf(L.ts_w2v_posconv_workspace_bytes(b, t, c, self.kpos), dtype=torch.uint8)
        hp = torch.empty_like(h)
        if self.d2v:
            # pos = (GELU . LayerNorm_no_affine . conv)^n (h); the encoder's LayerNorm below takes h as its residual: LN(pos + h)
            pos = h
            for w_taps, bias in self.pos_stack:
                _lib.check(L.ts_w2v_groupconv_fwd(pos.data_ptr(), b, t, c, w_taps.data_ptr(), bias.data_ptr(), self.kpos, self.groups, self.prec,
                                                  hp.data_ptr(), ws.data_ptr(), stream), "ts_w2v_groupconv_fwd")
                pos, _ = self._ln(L, stream, hp, self.unit_ln, act=1, eps=1e-5, want_op=False)
            hp, pos_res = pos, h
        else:
            pos_res = None                             # the wav2vec2 / hubert launch adds its input itself: hp = h + gelu(conv(h) + b)
            _lib.check(L.ts_w2v_posconv_fwd(h.data_ptr(), b, t, c, self.pos_w.data_ptr(), self.pos_b.data_ptr(), self.kpos, self.groups,
                                            self.prec, hp.data_ptr(), None, ws.data_ptr(), stream), "ts_w2v_posconv_fwd")
        del ws
        att_ws = self._buf(L.ts_w2v_attention_workspace_bytes(b, t, self.heads, self.prec), dtype=torch.uint8)

        def attention(x_op):
            _, qkv_op = self._linear(L, stream, x_op, lw["wqkv"], lw["bqkv"], want_op=True)
            ctx_op = self._buf(b, t, c, dtype=torch.bfloat16 if self.prec else torch.float32)
            _lib.check(L.ts_w2v_attention_fwd(qkv_op.data_ptr(), b, t, c, self.heads, self._ptr(key_len), self.prec, ctx_op.data_ptr(),
                                              att_ws.data_ptr(), stream), "ts_w2v_attention_fwd")
            return ctx_op

        if self.stable_ln:
            # pre-LN family: h += attn(LN(h)); h += ffn(LN(h)); one LayerNorm after the last layer
            assert pos_res is None
            h = hp
            for lw in self.layers:
                _, x_op = self._ln(L, stream, h, lw["ln1"], want_f32=False)
                self._linear(L, stream, attention(x_op), lw["wo"], lw["bo"], into=h)        # h += attn W^T (+ bias in the epilogue)
                _, x_op = self._ln(L, stream, h, lw["ln2"], want_f32=False)
                _, f1_op = self._linear(L, stream, x_op, lw["w1"], lw["b1"], act=1, want_op=True)
                self._linear(L, stream, f1_op, lw["w2"], lw["b2"], into=h)
            h, _ = self._ln(L, stream, h, self.enc_ln, want_op=False)
            return self._adapter(L, stream, h) if self.adapter else h
        # post-LN family: LayerNorm before the layers, after each residual add inside them
        h, h_op = self._ln(L, stream, hp, self.enc_ln, res=pos_res)
        for lw in self.layers:
            # the projections accumulate into the residual stream inside the GEMM; their biases ride in the LayerNorm launch
            self._linear(L, stream, attention(h_op), lw["wo"], None, into=h)
            h, h_op = self._ln(L, stream, h, lw["ln1"], xbias=lw["bo"])
            _, f1_op = self._linear(L, stream, h_op, lw["w1"], lw["b1"], act=1, want_op=True)
            self._linear(L, stream, f1_op, lw["w2"], None, into=h)
            h, h_op = self._ln(L, stream, h, lw["ln2"], xbias=lw["b2"])
        return self._adapter(L, stream, h) if self.adapter else h

    def _adapter(self, L, stream, h: torch.Tensor) -> torch.Tensor:
        """Wav2Vec2Adapter.forward (eval): [B, T, C] -> [B, T'', output_hidden_size]."""
        if self.ad_proj is not None:
            w, bias, ln = self.ad_proj
            b0, t0, c0 = h.shape
            y = self._buf(b0, t0, w.shape[0])
            _lib.check(L.ts_w2v_linear_fwd(h.data_ptr(), c0, w.data_ptr(), bias.data_ptr(), None, w.shape[0], y.data_ptr(), w.shape[0], None,
                                           b0 * t0, w.shape[0], c0, 0, 0, None, stream), "ts_w2v_linear_fwd")       # f32 operands (precision 0)
            h, _ = self._ln(L, stream, y, ln, want_op=False, eps=1e-5)
        b, t, c = h.shape
        for w, bias in self.ad_layers:
            t_pad = t + 2                                          # Conv1d(padding = 1)
            t_out = (t_pad - self.ad_k) // self.ad_s + 1
            if t_out < 1:
                raise RuntimeError("wav2vec2: input too short for the adapter layers")
            hp = self._buf(b, t_pad, c)
            _lib.check(L.ts_w2v_pad_rows(h.data_ptr(), hp.data_ptr(), b, t, t_pad, 1, c, 0, stream), "ts_w2v_pad_rows")
            y = self._buf(b, t_out, 2 * c)
            _lib.check(L.ts_w2v_conv_fwd(hp.data_ptr(), b, t_pad, c, w.data_ptr(), bias.data_ptr(), 2 * c, self.ad_k, self.ad_s, 0, 0, y.data_ptr(),
                                         None, None, stream), "ts_w2v_conv_fwd")
            h = self._buf(b, t_out, c)
            _lib.check(L.ts_w2v_glu_fwd(y.data_ptr(), b * t_out, c, h.data_ptr(), None, stream), "ts_w2v_glu_fwd")
            t = t_out
        return h


class HuggingFaceEncoderAdapt(nn.Module):
    """Same constructor, attributes and return convention as the reference's `_HuggingFaceEncoderAdapt`
    (huggingface/compatibility.py:23-42): `(audio [B, n], lengths) -> (features [B, C, T'], lengths')`."""

    def __init__(self, encoder, mask_input: bool = False, precision: str = "bf16", train_precision: str = "fp32"):
        if train_precision not in ("fp32", "bf16"):
            raise ValueError(f"train_precision must be 'fp32' or 'bf16', got {train_precision!r}")
        super().__init__()
        self.precision = precision
        # arithmetic of the TRAINABLE part in train mode (huggingface/train.py): "fp32" = the reference's, "bf16" = mixed precision (bf16 operands in the
        # linear layers' three products, everything else f32) -- the reference under Lightning's precision="bf16-mixed"
        self.train_precision = train_precision
        self.original_encoder = encoder
        if hasattr(self.original_encoder, "freeze_feature_encoder"):
            self.original_encoder.freeze_feature_encoder()
        self.mask_input = mask_input
        self._cache = _PackedCache()
        _check_config(encoder.config)

    def _plan(self, device) -> Wav2Vec2Plan:
        params = list(self.original_encoder.parameters())
        return self._cache.get(params, lambda: Wav2Vec2Plan(self.original_encoder.config, self.original_encoder.state_dict(), device, self.precision))

    def _plan_frozen(self, device) -> Wav2Vec2Plan:
        """The plan the training path runs the frozen conv feature extractor on: built from and keyed on the feature extractor's parameters
        only, so an optimizer step on the transformer re-packs nothing and no transformer weight gets a device copy it never uses.  It runs
        at `self.precision` -- with the default "bf16" the FROZEN convolutions multiply bf16 operands (f32 accumulation) while everything
        trainable behind them is f32; pass precision="fp32" for f32 arithmetic throughout (what the parity tests use)."""
        if not hasattr(self, "_fe_cache"):
            self._fe_cache = _PackedCache()
        fe = self.original_encoder.feature_extractor
        params = list(fe.parameters())
        sd = {"feature_extractor." + k: v for k, v in fe.state_dict().items()}
        return self._fe_cache.get(params, lambda: Wav2Vec2Plan(self.original_encoder.config, sd, device, self.precision, feature_extractor_only=True))

    def forward(self, audio: torch.Tensor, audio_lengths: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        _t.require_gpu(audio, "wav2vec2 encoder")
        x = audio.to(torch.float32).contiguous()
        if self.training:
            # fine-tuning (module.py:102-127 through this adapter): transformers' training-mode forward -- frozen conv feature extractor, time
            # masking, dropouts, LayerDrop -- as a chain of autograd nodes on the HIP kernels (huggingface/train.py), f32 arithmetic
            from .train import train_forward
            h = train_forward(self, x, audio_lengths if self.mask_input else None)
        else:
            h = self._plan(x.device).forward(x, audio_lengths if self.mask_input else None)
        cfg = self.original_encoder.config
        n_ad = int(cfg.num_adapter_layers) if getattr(cfg, "add_adapter", False) else 0
        return h.transpose(-1, -2), feat_extract_output_lengths(cfg.conv_kernel, cfg.conv_stride, audio_lengths, n_ad, int(getattr(cfg, "adapter_stride", 2)))
