"""Training-mode encoder ops (fine-tuning with the encoder unfrozen): one torch.autograd.Function per reference op, each
forward AND backward a call into csrc/train_enc.hip / train_extra.hip.  PyTorch only chains the Functions; no ATen compute op
touches an activation.

Activation tensors are [B, C, T] views of pitched rows ([B, C, P], P = T rounded up to 192 frames, 32-byte aligned) in ONE of
two element types, chosen with `set_activation_dtype`:
  * "fp32" (default): the reference's arithmetic -- what the parity tests against the reference's autograd use;
  * "bf16": mixed precision (the reference under Lightning's precision="bf16-mixed"): activations and their gradients are
    stored as bf16, every kernel computes in f32, the pointwise GEMMs run on bf16 operands with f32 accumulation; parameters,
    parameter gradients, BatchNorm statistics and the logits stay f32.
`to_act` / `from_act` are the boundary Functions (reference-layout f32 tensor <-> activation rows)."""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib

_ACT_DTYPE = torch.float32


def set_activation_dtype(name: str) -> None:
    """"fp32" or "bf16": the element type of every activation the training path allocates from now on."""
    global _ACT_DTYPE
    if name not in ("fp32", "bf16"):
        raise ValueError(f"activation dtype must be 'fp32' or 'bf16', got {name!r}")
    _ACT_DTYPE = torch.bfloat16 if name == "bf16" else torch.float32


def activation_dtype() -> torch.dtype:
    return _ACT_DTYPE


_DET_WS = {}


def set_deterministic(on: bool, device=None, workspace_floats: int = 8 << 20) -> None:
    """Deterministic gradients (the notion of torch.use_deterministic_algorithms for this library's training kernels): the depthwise backward
    kernels then leave their per-workgroup sums of the weight / BatchNorm-parameter gradients in a workspace and a second launch adds them in a
    fixed order, instead of float atomics -- every other kernel of the step already sums in a fixed order (split-K partials + ordered reduce,
    per-tile statistics, one workgroup per channel).  A training run started from the same seeds is then bit-reproducible, at one extra tiny
    launch per depthwise layer (measured on QuartzNet15x5, local 32 x 10 s: 9.05 against 8.57 ms per step, +5.5 %; profiles/round6_c4_pointwise.md section 7).  Process-wide (ts_train_set_deterministic); the workspace (32 MiB by default) lives until the mode is switched off."""
    if on:
        dev = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
        ws = torch.empty(int(workspace_floats), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().ts_train_set_deterministic(ws.data_ptr(), ws.numel()), "ts_train_set_deterministic")
        _DET_WS["ws"] = ws
    else:
        _lib.check(_lib.lib().ts_train_set_deterministic(None, 0), "ts_train_set_deterministic")
        _DET_WS.clear()


def set_gemm_precision(precision: str) -> None:
    """Round-1 name of the mixed-precision switch: "bf16" now selects bf16 activations (and with them bf16 GEMM operands)."""
    set_activation_dtype(precision)


def _s(t: Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def _nonce(t: Tensor):
    """Device replay counter for the dropout seed -- only while the launch is being captured into a hipGraph (whose by-value seed is
    frozen); eager launches draw a fresh seed per call and pass NULL, so their masks are a pure function of that seed."""
    if not torch.cuda.is_current_stream_capturing():
        return None
    from .rng import replay_nonce
    return replay_nonce(t.device).data_ptr()


def _code(t: Tensor) -> int:
    return 1 if t.dtype == torch.bfloat16 else 0


PITCH_QUANTUM = 192      # frames; the pointwise-only mode of the inference kernel (forward / data-gradient GEMMs) works in 96- and
                         # 192-frame tiles and reads / writes whole tiles: rows are padded so that every tile lies inside its row


def row_pitch(t: int) -> int:
    # whole 192-frame tiles + the 64-frame granule the 96-frame tiling's last stage copy reaches past the final tile start
    return (t + PITCH_QUANTUM - 1) // PITCH_QUANTUM * PITCH_QUANTUM + 64


def alloc(b: int, c: int, t: int, device, dtype=None) -> Tensor:
    """Uninitialised activation [b, c, t] on pitched rows."""
    return torch.empty(b, c, row_pitch(t), dtype=dtype or _ACT_DTYPE, device=device)[:, :, :t]


def alloc_like(x: Tensor) -> Tensor:
    return alloc(x.shape[0], x.shape[1], x.shape[2], x.device, x.dtype)


def is_act(x: Tensor) -> bool:
    # the pitch must be THE pitch `alloc` gives this length: kernels take one pitch for a tensor and its gradient / its
    # same-shape output, so a foreign row layout (e.g. the front end's feature rows, padded to its own tile) is re-packed
    return (x.is_cuda and x.dim() == 3 and x.dtype in (torch.float32, torch.bfloat16) and x.stride(2) == 1
            and x.stride(1) == row_pitch(x.shape[2]) and x.stride(0) == x.shape[1] * x.stride(1) and x.data_ptr() % 32 == 0)


def _pitch(x: Tensor) -> int:
    return x.stride(1)


def _import(x: Tensor, dtype) -> Tensor:
    """Any [B, C, T] tensor -> activation rows of `dtype` (a copy unless it already is one)."""
    if not x.is_cuda:
        raise RuntimeError("thunder_speech_amd training ops run on the GPU only (no CPU fallback)")
    if is_act(x) and x.dtype == dtype:
        return x
    if is_act(x):                                            # the other element type: through f32
        x = _export(x)
    src = x.to(torch.float32).contiguous()
    b, c, t = src.shape
    out = alloc(b, c, t, src.device, dtype)
    st = _lib.lib().ts_train_act_import(src.data_ptr(), out.data_ptr(), b * c, t, _pitch(out), _code(out), _s(out))
    _lib.check(st, "ts_train_act_import")
    return out


def _export(x: Tensor) -> Tensor:
    """Activation rows -> contiguous f32 [B, C, T]."""
    b, c, t = x.shape
    out = torch.empty(b, c, t, dtype=torch.float32, device=x.device)
    st = _lib.lib().ts_train_act_export(x.data_ptr(), out.data_ptr(), b * c, t, _pitch(x), _code(x), _s(x))
    _lib.check(st, "ts_train_act_export")
    return out


class _ToAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src_dtype = x.dtype
        return _import(x, dtype)

    @staticmethod
    def backward(ctx, dy):
        return _export(_import(dy, dy.dtype if is_act(dy) else torch.float32)).to(ctx.src_dtype), None


class _FromAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.dtype = x.dtype
        return _export(x)

    @staticmethod
    def backward(ctx, dy):
        return _import(dy, ctx.dtype)


def to_act(x: Tensor) -> Tensor:
    """Boundary into the training path: activation rows of the current activation dtype (identity if x already is one)."""
    if is_act(x) and x.dtype == _ACT_DTYPE:
        return x
    return _ToAct.apply(x, _ACT_DTYPE)


def from_act(x: Tensor) -> Tensor:
    """Boundary out of the training path: contiguous f32 [B, C, T] (autograd flows back into the activation rows)."""
    return _FromAct.apply(x)


def _g(dy: Tensor, like: Tensor) -> Tensor:
    """An incoming gradient as activation rows of `like`'s element type (test cotangents arrive as plain f32 tensors)."""
    return _import(dy, like.dtype)


# Parameter-gradient buffers.  parallel.GradientSync registers, per parameter, the view of its flat bucket buffer the gradient
# has to end up in (and itself: it knows whether the buffer is still all zeros); the backward kernels then write there directly and
# autograd adopts the returned tensor as `.grad` (no per-parameter accumulate kernel, no copy into the bucket).
_GRAD_VIEWS = {}


def grad_out_ex(param, shape, zeroed: bool = False):
    """(tensor, is_bucket_view): f32 tensor of `shape` for the gradient of `param` -- its registered flat-buffer view when it has one and
    `.grad` is unset (first -- and in the fine-tuning loop only -- contribution of the step), else a fresh tensor.  `zeroed`: the kernel
    about to receive it ACCUMULATES, so the tensor must hold zeros -- free for a bucket view (GradientSync.zero_grad cleared the whole
    buffer with one memset), one fill launch otherwise."""
    ent = _GRAD_VIEWS.get(id(param)) if param is not None else None
    if ent is not None and param.grad is None and ent[1].claim(param):
        # (claim: a parameter that contributes twice in one backward -- tied weights, a module called twice -- gets the bucket view for
        # its first contribution only; the second one takes a fresh tensor and autograd adds the two)
        view, owner = ent
        out = view.view(shape)
        if zeroed and not owner.is_clean():
            out.zero_()
        return out, True
    dev = ent[0].device if ent is not None else param.device
    return (torch.zeros if zeroed else torch.empty)(shape, dtype=torch.float32, device=dev), False


def grad_out(param, shape, zeroed: bool = False) -> Tensor:
    return grad_out_ex(param, shape, zeroed)[0]


# bf16 operand copies of the weights (mixed precision).  One per parameter, reused while the parameter's version counter stands;
# optim.FusedAdamW writes the copy inside its update kernel and re-stamps it, so a fine-tuning step launches no cast at all.
_BF16_SHADOW = {}


def bf16_shadow(param: Tensor, create: bool = False):
    """[bf16 copy [C_out, rest], version it matches, weakref to its parameter] or None."""
    import weakref
    ent = _BF16_SHADOW.get(id(param))
    if ent is not None and (ent[2]() is not param or ent[0].numel() != param.numel() or ent[0].device != param.device):
        ent = None                               # the id was recycled by another tensor, or the parameter moved
        _BF16_SHADOW.pop(id(param), None)
    if ent is None and create:
        if len(_BF16_SHADOW) > 4096:             # entries of parameters that no longer exist
            for k in [k for k, e in _BF16_SHADOW.items() if e[2]() is None]:
                del _BF16_SHADOW[k]
        ent = _BF16_SHADOW[id(param)] = [torch.empty(param.shape[0], param.numel() // max(param.shape[0], 1), dtype=torch.bfloat16,
                                                     device=param.device), -1, weakref.ref(param)]
    return ent


def _w_bf16(w2: Tensor, param: Tensor = None) -> Tensor:
    ent = bf16_shadow(param, create=True) if (param is not None and param.dim() >= 2) else None
    if ent is not None and ent[1] == param._version:
        return ent[0]
    y = ent[0] if ent is not None else torch.empty(w2.shape, dtype=torch.bfloat16, device=w2.device)
    _lib.check(_lib.lib().ts_train_cast_bf16(w2.data_ptr(), y.data_ptr(), w2.numel(), _s(w2)), "ts_train_cast_bf16")
    if ent is not None:
        ent[1] = param._version
    return y


# bf16 pointwise convolutions: forward and data gradient on the inference kernel's pointwise-only mode (ts_tcs_subblock_fwd), weight
# gradient on csrc/train_gemm.hip; set_pointwise_backend("gemm_f32") routes all three to this library's general GEMM (csrc/gemm_f32.hip, bf16
# operands) instead -- what the f32 path always uses; an A/B and fallback switch.  ("rocblas", the name from the rounds when that path was the
# vendor's strided-batched call, is still accepted: no vendor library has been linked since round 4.)
_OWN_GEMM = True


def set_pointwise_backend(name: str) -> None:
    global _OWN_GEMM
    if name not in ("mfma", "gemm_f32", "rocblas"):
        raise ValueError("pointwise backend must be 'mfma' or 'gemm_f32'")
    _OWN_GEMM = name == "mfma"


# Weights in MFMA B-fragment order (W for the forward product, W^T for the data gradient), one pair per parameter, reused while the
# parameter's version counter stands.  optim.FusedAdamW re-packs every registered pair in ONE launch after its update
# (refresh_pw_frags), so a fine-tuning step launches no per-layer packing.
_PW_FRAGS = {}


def _frag_shapes(c_out: int, c_in: int):
    r = lambda x, m: (x + m - 1) // m * m
    return (r(c_out, 32) // 32, r(c_in, 64) // 16, 64, 8), (r(c_in, 32) // 32, r(c_out, 64) // 16, 64, 8)


def _pack_rows(entries):
    """ONE launch packing the (param, w2 f32 [c_out, c_in], fwd, bwd) entries."""
    dev = entries[0][1].device
    if torch.cuda.is_current_stream_capturing():
        # the pointer table travels host -> device from a temporary pinned buffer: captured, every replay would re-read freed host memory
        raise RuntimeError("weight fragments cannot be (re)packed while a hipGraph is being captured: run an eager pass first "
                           "(GraphedTrainStep does; warmup >= 1)")
    rows = [[w2.data_ptr(), f.data_ptr(), bk.data_ptr(), w2.shape[0], w2.shape[1]] for _, w2, f, bk in entries]
    groups = max(f.numel() // 8 + bk.numel() // 8 for _, _, f, bk in entries)
    table = torch.tensor(rows, dtype=torch.int64, pin_memory=True).to(dev, non_blocking=True)
    _lib.check(_lib.lib().ts_train_pack_pw_multi(table.data_ptr(), len(rows), groups, torch.cuda.current_stream(dev).cuda_stream), "ts_train_pack_pw_multi")


def pw_frags(param: Tensor, w2: Tensor):
    """(forward fragments, backward fragments) of the 1x1-conv weight `param` (w2 = its f32 [c_out, c_in] view)."""
    import weakref
    ent = _PW_FRAGS.get(id(param))
    if ent is not None and (ent[3]() is not param or ent[0].device != param.device):
        ent = None
    if ent is None:
        if len(_PW_FRAGS) > 4096:
            for k in [k for k, e in _PW_FRAGS.items() if e[3]() is None]:
                del _PW_FRAGS[k]
        sf, sb = _frag_shapes(w2.shape[0], w2.shape[1])
        ent = _PW_FRAGS[id(param)] = [torch.empty(sf, dtype=torch.bfloat16, device=param.device),
                                      torch.empty(sb, dtype=torch.bfloat16, device=param.device), -1, weakref.ref(param)]
    if ent[2] != param._version:
        _pack_rows([(param, w2, ent[0], ent[1])])
        ent[2] = param._version
    return ent[0], ent[1]


def refresh_pw_frags(params) -> None:
    """Re-pack, in one launch, the fragment pairs of those `params` that have one (called by FusedAdamW after its update)."""
    todo = []
    for p in params:
        ent = _PW_FRAGS.get(id(p))
        if ent is not None and ent[3]() is p and ent[2] != p._version and p.dtype == torch.float32 and p.is_contiguous():
            todo.append((p, p.detach().view(p.shape[0], -1), ent[0], ent[1]))
    if todo:
        _pack_rows(todo)
        for p, _, _, _ in todo:
            _PW_FRAGS[id(p)][2] = p._version


def refresh_weight_copies(params) -> None:
    """Bring every derived copy of `params` the training path keeps (MFMA fragments, plain bf16 operands) up to date with the
    parameters' current versions.  The eager path does this lazily inside each forward; a step replayed from a hipGraph cannot -- its
    launches read the copies' buffers as captured -- so train_graph.GraphedTrainStep calls this after every optimizer step (a no-op
    after FusedAdamW, which refreshes them itself; what keeps any OTHER optimizer correct)."""
    refresh_pw_frags(params)
    for p in params:
        ent = _BF16_SHADOW.get(id(p))
        if ent is not None and ent[2]() is p and ent[1] != p._version:
            _w_bf16(p.detach().to(torch.float32).contiguous().view(p.shape[0], -1), p)


_ZERO_BIAS = {}


def _zero_bias(n: int, device) -> Tensor:
    key = str(torch.device(device))
    z = _ZERO_BIAS.get(key)
    if z is None or z.numel() < n:
        z = _ZERO_BIAS[key] = torch.zeros(max(4096, (n + 31) // 32 * 32), dtype=torch.float32, device=device)
    return z


def _tcs_ok(bf: bool, k: int) -> bool:
    return _OWN_GEMM and bf and k % 64 == 0


def _tcs_pointwise(x: Tensor, frags: Tensor, y: Tensor, lens: Tensor, n_out: int, stats: Tensor = None) -> None:
    """y[b] = Wn . x[b] on the inference kernel's pointwise-only mode: x bf16 rows [B, K, T], frags = B-fragments of Wn [n_out, K],
    y bf16 rows [B, n_out, T] -- or f32 rows (the decoder's logits: the kernel's fp32-output mode, what inference uses for them; through the f32
    GEMM a 29-row product costs 113 us of mostly padding, here 20).  Frames beyond T inside the row pitch are read and written as scratch
    (every frame is independent)."""
    import ctypes as C
    b, k, t = x.shape
    d = _lib.TcsDesc()
    d.batch, d.c_in, d.c_out, d.t_in, d.t_out, d.pitch_in, d.pitch_out = b, k, n_out, t, t, _pitch(x), _pitch(y)
    d.kernel, d.stride, d.dilation, d.padding, d.depthwise, d.relu, d.out_fp32 = 1, 1, 1, 0, 0, 0, int(y.dtype == torch.float32)
    # no TS_TCS_IN_TAILZERO: with `lens` = the full length the generic kernel (4 producer + 4 consumer waves, 64 / 128-frame tiles) runs, which
    # at these sizes (32 x 501 frames) is 2 us per launch faster than the split kernel the flag would select (15.5 vs 17.5 us at 512^2,
    # 8.8 vs 10.8 at 256^2; round-2 measurement, tools/diag/pw_tile_bench.py times the kernel now) -- the split kernel's 96-frame tiles and 12 waves pay off at inference batch sizes
    d.flags = 0
    d.pw_w, d.bias = frags.data_ptr(), _zero_bias(n_out, x.device).data_ptr()
    if stats is not None:                # per-tile BatchNorm statistics of y out of the launch's epilogue (ts_tcs_desc.stats)
        d.stats = stats.data_ptr()
    st = _lib.lib().ts_tcs_subblock_fwd(C.byref(d), x.data_ptr(), lens.data_ptr(), None, None, y.data_ptr(), _s(x))
    _lib.check(st, "ts_tcs_subblock_fwd")


# Deferred split-K reduction of the pointwise weight gradients.  Eagerly, every layer's partial products are summed onto its gradient right
# behind the product (GradientSync's hooks may send the bucket the moment autograd hands the gradient over).  A step replayed from hipGraphs
# runs no hook, so train_graph.GraphedTrainStep opens `deferred_wgrad()` around each piece of its backward pass: the layers only REGISTER their
# weight-gradient product (operands kept alive in the list below) and `flush_wgrad()` runs all of them at the end of the piece -- the split-K
# products of up to 32 layers per launch (ts_train_pwconv_wgrad_multi), then ONE summation launch per 64 layers (ts_train_wgrad_reduce_multi).
# Nothing reads a weight gradient before the optimizer, and every launch saved is ~5 us of latency, ramp and drain at this size.
_WGRAD_PENDING = None
MAX_PENDING_WGRAD = 32      # = the layers one grouped launch takes (ts_train_pwconv_wgrad_multi): flushing there costs no extra product launch
GROUP_WGRAD = True          # False: inside deferred_wgrad() every layer still launches its own product at once and only the summation waits (A/B)


class deferred_wgrad:
    """Context: park the split-K partials of every bf16 pointwise weight gradient computed inside; flush_wgrad() (called on exit too) sums them."""

    def __enter__(self):
        global _WGRAD_PENDING
        self._outer = _WGRAD_PENDING
        if _WGRAD_PENDING is None:
            _WGRAD_PENDING = []
        return self

    def __exit__(self, *exc):
        global _WGRAD_PENDING
        try:
            if exc[0] is None:
                flush_wgrad()
        finally:
            if self._outer is None:
                _WGRAD_PENDING = None
        return False


def flush_wgrad() -> int:
    """Sum every parked partial onto its gradient (ts_train_wgrad_reduce_multi); returns the number of layers.  All on the current stream."""
    import ctypes as C
    pend = _WGRAD_PENDING
    if not pend:
        return 0
    n = len(pend)
    stream = _s(pend[0][1])
    todo = [e for e in pend if e[3] is not None]                 # products not launched yet
    items = (_lib.WgradItem * max(len(todo), 1))()
    for it, (ws, dw, k, dv, u, len_u) in zip(items, todo):
        b, c_out, t = dv.shape
        it.dv, it.u, it.len_u, it.workspace = dv.data_ptr(), u.data_ptr(), (len_u.data_ptr() if len_u is not None else None), ws.data_ptr()
        it.batch, it.c_in, it.c_out, it.t, it.pitch_u, it.pitch_v = b, u.shape[1], c_out, t, _pitch(u), _pitch(dv)
    st, st2 = (_lib.lib().ts_train_pwconv_wgrad_multi(items, len(todo), stream) if todo else 0), 0
    if st == 0:
        parts = (C.c_void_p * n)(*[e[0].data_ptr() for e in pend])
        dws = (C.c_void_p * n)(*[e[1].data_ptr() for e in pend])
        sizes = (C.c_int64 * n)(*[e[1].numel() for e in pend])
        nparts = (C.c_int32 * n)(*[e[2] for e in pend])
        st2 = _lib.lib().ts_train_wgrad_reduce_multi(parts, dws, sizes, nparts, n, stream)
    pend.clear()
    _lib.check(st, "ts_train_pwconv_wgrad_multi")
    _lib.check(st2, "ts_train_wgrad_reduce_multi")
    return n


def _wgrad(dv: Tensor, u: Tensor, dw: Tensor, len_u: Tensor = None, defer: bool = False) -> None:
    """dw += sum_b dv[b] . mask(u[b], len_u)^T (csrc/train_gemm.hip); dv [B, c_out, T], u [B, c_in, T] bf16 rows, dw f32 [c_out, c_in].
    `defer` (only inside deferred_wgrad()): leave the partials, flush_wgrad() sums them."""
    L = _lib.lib()
    b, c_out, t = dv.shape
    c_in = u.shape[1]
    if defer and GROUP_WGRAD and _WGRAD_PENDING is not None and (c_out * c_in) % 4 == 0:
        n_parts = L.ts_train_pwconv_wgrad_multi_parts(b, c_in, c_out)               # the grouped launch splits a layer over fewer clip groups
        ws = torch.empty(n_parts * c_out * c_in, dtype=torch.float32, device=dv.device)
        _WGRAD_PENDING.append((ws, dw, n_parts, dv, u, len_u))                      # operands stay alive until flush_wgrad()
        if len(_WGRAD_PENDING) >= MAX_PENDING_WGRAD:                               # bounds the memory parked operands hold (2 tensors per layer)
            flush_wgrad()
        return
    n_ws = L.ts_train_pwconv_wgrad_workspace(b, c_in, c_out)
    ws = torch.empty(n_ws, dtype=torch.float32, device=dv.device)
    if defer and _WGRAD_PENDING is not None and (c_out * c_in) % 4 == 0:
        if GROUP_WGRAD:
            _WGRAD_PENDING.append((ws, dw, n_ws // (c_out * c_in), dv, u, len_u))   # operands stay alive until flush_wgrad()
            return
        _lib.check(L.ts_train_pwconv_wgrad_mfma(dv.data_ptr(), u.data_ptr(), len_u.data_ptr() if len_u is not None else None, None, ws.data_ptr(),
                                                b, c_in, c_out, t, _pitch(u), _pitch(dv), _s(dv)), "ts_train_pwconv_wgrad_mfma")
        _WGRAD_PENDING.append((ws, dw, n_ws // (c_out * c_in), None, None, None))
        return
    _lib.check(L.ts_train_pwconv_wgrad_mfma(dv.data_ptr(), u.data_ptr(), len_u.data_ptr() if len_u is not None else None,
                                            dw.data_ptr(), ws.data_ptr(), b, c_in, c_out, t, _pitch(u), _pitch(dv), _s(dv)),
               "ts_train_pwconv_wgrad_mfma")


_LENS = {}


def _full_lengths(b: int, t: int, device) -> Tensor:
    key = (b, t, str(torch.device(device)))
    if key not in _LENS:
        if torch.cuda.is_current_stream_capturing():
            # a tensor born during capture lives in the graph's private pool: never cache it for eager callers
            return torch.full((b,), t, dtype=torch.int32, device=device)
        if len(_LENS) > 256:
            _LENS.clear()
        _LENS[key] = torch.full((b,), t, dtype=torch.int32, device=device)
    return _LENS[key]


def _pw_masks_inside(u_dtype, c_in: int, c_out: int) -> bool:
    """True when both directions of this 1x1 conv run on kernels that apply the MaskedConv1d input mask themselves (the generic
    pointwise kernel masks by length; the weight-gradient kernel takes len_u), so the masked copy of the input need not be made."""
    bf = u_dtype == torch.bfloat16
    return _tcs_ok(bf, c_in) and _tcs_ok(bf, c_out) and c_in % 8 == 0


def tile_stats_buffer(b: int, c_out: int, t: int, device) -> Tensor:
    """f32 [c_out, batch * n_tiles, 2] for ts_tcs_desc.stats of the pointwise launch over b x t frames (every entry is written by the launch)."""
    tt = _lib.lib().ts_tcs_pointwise_tile_frames(b, c_out, t)
    if tt <= 0:
        raise RuntimeError("ts_tcs_pointwise_tile_frames failed")
    return torch.empty(c_out, b * ((t + tt - 1) // tt), 2, dtype=torch.float32, device=device)


def _pw_fwd(u: Tensor, param: Tensor, w2: Tensor, f32_out: bool = False, lens: Tensor = None, stats: Tensor = None) -> Tensor:
    """v[b] = W . u[b] for activation rows u [B, c_in, T]; w2 = f32 [c_out, c_in] view of `param`.  `lens` (only with
    _pw_masks_inside): u is NOT masked yet, frames >= lens[b] count as zero; otherwise u is masked already."""
    b, c_in, t = u.shape
    c_out = w2.shape[0]
    bf = u.dtype == torch.bfloat16
    v = alloc(b, c_out, t, u.device, torch.float32 if (f32_out or not bf) else torch.bfloat16)
    if _tcs_ok(bf, c_in):
        _tcs_pointwise(u, pw_frags(param, w2)[0], v, lens if lens is not None else _full_lengths(b, t, u.device), c_out, stats)
    else:
        if stats is not None:
            raise RuntimeError("_pw_fwd: per-tile statistics come out of the matrix-core pointwise kernel only")
        if lens is not None:
            raise RuntimeError("_pw_fwd: an unmasked input needs the kernels that mask inside")
        wk = _w_bf16(w2, param) if bf else w2
        prec = 0 if not bf else (1 if f32_out else 2)
        st = _lib.lib().ts_train_pwconv_fwd(u.data_ptr(), wk.data_ptr(), v.data_ptr(), b, c_in, c_out, t, _pitch(u), _pitch(v), prec, _s(v))
        _lib.check(st, "ts_train_pwconv_fwd")
    return v


def _pw_bwd(dv: Tensor, u: Tensor, param: Tensor, w2: Tensor, len_u: Tensor = None):
    """(du, dw) = (W^T . dv[b], sum_b dv[b] . u[b]^T); dw f32 [c_out, c_in] (the bucket view of `param` when it has one).  len_u: u is
    the UNMASKED input (see _pw_fwd); du is then the gradient w.r.t. the masked input -- the caller still owes the mask's backward."""
    b, c_in, t = u.shape
    c_out = w2.shape[0]
    bf = u.dtype == torch.bfloat16
    du = alloc_like(u)
    if _tcs_ok(bf, c_out) and c_in % 8 == 0:
        _tcs_pointwise(dv, pw_frags(param, w2)[1], du, _full_lengths(b, t, u.device), c_in)
        if not param.requires_grad:
            # frozen weight (the first phase of the reference's fine-tuning schedule freezes the convolutions but still backpropagates
            # through them to the BatchNorm parameters, callbacks.py): the data gradient is all that is needed
            return du, None
        dw, is_view = grad_out_ex(param, (c_out, c_in), zeroed=True)
        # only a bucket view may wait for the end of the piece: a fresh tensor (a second contribution to the same parameter) is added to
        # the first by autograd right away
        _wgrad(dv, u, dw, len_u, defer=is_view)
        return du, dw
    if len_u is not None:
        raise RuntimeError("_pw_bwd: an unmasked input needs the kernels that mask inside")
    wk = _w_bf16(w2, param) if bf else w2
    dw = grad_out(param, (c_out, c_in))
    ws = torch.empty(b * c_out * c_in, dtype=torch.float32, device=u.device)
    st = _lib.lib().ts_train_pwconv_bwd(dv.data_ptr(), u.data_ptr(), wk.data_ptr(), du.data_ptr(), dw.data_ptr(), ws.data_ptr(), b, c_in,
                                        c_out, t, _pitch(u), _pitch(dv), 2 if bf else 0, _s(du))
    _lib.check(st, "ts_train_pwconv_bwd")
    return du, dw


# Gradients of a block output that feeds the next block's main AND residual branch: Fork.backward parks the pair here instead of adding them
# (ts_train_add, 9 us x 15 per QuartzNet15x5 step) when the tensor came out of a one-launch block tail, whose backward kernel reads both
# (ts_train_bn2_chan_bwd's dout2).  Entries live from one autograd node to the next of the same backward pass; GradientSync clears leftovers.
DEFER_FORK_ADD = True
_PENDING_ADD = {}
_PENDING_CHECK_QUEUED = False


def _check_pending_add() -> None:
    """End-of-backward callback (queued by Fork.backward the first time it parks a pair in a backward pass): every parked pair must have
    been taken by a BlockTail.backward.  A leftover means the block output had ANOTHER consumer besides the Fork (a forward hook, an
    auxiliary / intermediate-CTC loss, a retained clone): autograd then summed the first gradient with the other consumer's into a new
    buffer, the key missed, and the residual-branch gradient would be silently missing from the step -- fail instead."""
    global _PENDING_CHECK_QUEUED
    _PENDING_CHECK_QUEUED = False
    if _PENDING_ADD:
        n = len(_PENDING_ADD)
        _PENDING_ADD.clear()
        raise RuntimeError(f"train_ops.Fork: {n} parked gradient pair(s) were never consumed -- a block output that feeds the next block has a second "
                           "consumer (hook / auxiliary loss); set train_ops.DEFER_FORK_ADD = False for such graphs")


def take_pending_add(dout: Tensor):
    """(second gradient, its length mask) parked for `dout` by Fork.backward, or (None, None)."""
    ent = _PENDING_ADD.pop(dout.data_ptr(), None)
    if ent is None or ent[0] is not dout and ent[0].data_ptr() != dout.data_ptr():
        return None, None
    return ent[1], ent[2]


class Fork(torch.autograd.Function):
    """x -> (x, x) for a tensor with two consumers (a block input: main branch + residual branch).  The backward pass adds the two
    gradients with one kernel on activation rows; autograd's own accumulation (an ATen add) leaves the row layout, which costs a
    re-import on the way."""

    @staticmethod
    def forward(ctx, x, res_len=None):
        global _PENDING_CHECK_QUEUED
        _PENDING_CHECK_QUEUED = False  # a new forward pass: whatever an interrupted backward left behind is over (its callbacks never ran)
        ctx.res_len = res_len          # int32 lengths: the second output feeds a MaskedConv1d whose input mask's backward is applied HERE
        # x is the output of a one-launch block tail (block_tail tags it): its backward kernel adds the two gradients itself
        ctx.defer_add = bool(getattr(x, "_ts_tail_out", False)) and DEFER_FORK_ADD
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None or g2 is None:
            if g2 is not None and ctx.res_len is not None:
                return MaskTime.apply(g2, ctx.res_len), None
            return (g1 if g2 is None else g2), None
        dtype = g1.dtype if is_act(g1) else (g2.dtype if is_act(g2) else _ACT_DTYPE)
        g1, g2 = _import(g1, dtype), _import(g2, dtype)
        if ctx.defer_add and chan_fits(g1.shape[0], g1.shape[2], g1.dtype):
            # the consumer is BlockTail.backward of the previous block: hand it both gradients (keyed by the first one's address; the entry keeps
            # the tensors alive, so the address cannot be reused while it is pending)
            global _PENDING_CHECK_QUEUED
            _PENDING_ADD[g1.data_ptr()] = (g1, g2, ctx.res_len)
            if not _PENDING_CHECK_QUEUED:
                # once per backward pass: when the engine has run every node, nothing may be left parked (works for a plain
                # loss.backward() + optimizer.step() loop too, which never calls GradientSync.finish / zero_grad)
                torch.autograd.Variable._execution_engine.queue_callback(_check_pending_add)
                _PENDING_CHECK_QUEUED = True
            return g1, None
        out = alloc_like(g1)
        ln = ctx.res_len
        st = _lib.lib().ts_train_add(g1.data_ptr(), g2.data_ptr(), ln.data_ptr() if ln is not None else None, g1.shape[1], out.data_ptr(),
                                     g1.shape[0] * g1.shape[1], g1.shape[2], _pitch(g1), _code(g1), _s(g1))
        _lib.check(st, "ts_train_add")
        return out, None


class DepthwiseConv(torch.autograd.Function):
    """MaskedConv1d with groups = C: y = conv(mask(x, len_in)); backward masks dx the same way.  With `len_out` the output is
    zeroed from len_out[b] on (the re-masking the following MaskedConv1d would apply) and so is the incoming gradient."""

    @staticmethod
    def forward(ctx, x, w, len_in, k, stride, dil, pad, len_out=None):
        x = _import(x, x.dtype if is_act(x) else _ACT_DTYPE)
        w2 = w.detach().to(torch.float32).contiguous().view(w.shape[0], -1)
        b, c, t_in = x.shape
        t_out = (t_in + 2 * pad - dil * (k - 1) - 1) // stride + 1
        y = alloc(b, c, t_out, x.device, x.dtype)
        st = _lib.lib().ts_train_dwconv_fwd(x.data_ptr(), len_in.data_ptr(), len_out.data_ptr() if len_out is not None else None,
                                            w2.data_ptr(), y.data_ptr(), b, c, t_in, t_out, k, stride, dil, pad, _pitch(x), _pitch(y),
                                            _code(x), _s(x))
        _lib.check(st, "ts_train_dwconv_fwd")
        ctx.save_for_backward(x, w2, len_in)
        ctx.len_out, ctx.param = len_out, w
        ctx.geom = (k, stride, dil, pad, t_out, w.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2, len_in = ctx.saved_tensors
        k, stride, dil, pad, t_out, wshape = ctx.geom
        dy = _g(dy, x)
        b, c, t_in = x.shape
        dx = alloc_like(x)
        dw = grad_out(ctx.param, w2.shape, zeroed=True) if ctx.param.requires_grad else None      # frozen weight: data gradient only
        lo = ctx.len_out
        st = _lib.lib().ts_train_dwconv_bwd(dy.data_ptr(), x.data_ptr(), len_in.data_ptr(), lo.data_ptr() if lo is not None else None,
                                            w2.data_ptr(), dx.data_ptr(), dw.data_ptr() if dw is not None else None, b, c, t_in, t_out, k, stride,
                                            dil, pad, _pitch(x), _pitch(dy), _code(x), _s(x))
        _lib.check(st, "ts_train_dwconv_bwd")
        return dx, (dw.view(wshape) if dw is not None else None), None, None, None, None, None, None


class MaskTime(torch.autograd.Function):
    """Zero the frames >= length (the re-masking in front of every MaskedConv1d); the gradient is masked the same way."""

    @staticmethod
    def forward(ctx, x, lens):
        x = _import(x, x.dtype if is_act(x) else _ACT_DTYPE)
        b, c, t = x.shape
        y = alloc_like(x)
        st = _lib.lib().ts_train_mask_time(x.data_ptr(), lens.data_ptr(), y.data_ptr(), b, c, t, _pitch(x), _pitch(y), _code(x), _s(x))
        _lib.check(st, "ts_train_mask_time")
        ctx.save_for_backward(lens)
        ctx.dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        (lens,) = ctx.saved_tensors
        dy = _import(dy, ctx.dtype)
        b, c, t = dy.shape
        dx = alloc_like(dy)
        st = _lib.lib().ts_train_mask_time(dy.data_ptr(), lens.data_ptr(), dx.data_ptr(), b, c, t, _pitch(dy), _pitch(dx), _code(dy), _s(dy))
        _lib.check(st, "ts_train_mask_time")
        return dx, None


class PointwiseConv(torch.autograd.Function):
    """1x1 conv without bias on an already masked input: v[b] = W . u[b], du = W^T dv, dW = sum_b dv u^T.
    f32 activations: f32 GEMMs.  bf16 activations: bf16 operands, f32 accumulation; the result is bf16 too, except with
    `f32_out` (the decoder's logits feed the CTC kernel in f32)."""

    @staticmethod
    def forward(ctx, u, w, f32_out=False):
        u = _import(u, u.dtype if is_act(u) else _ACT_DTYPE)
        w2 = w.detach().to(torch.float32).contiguous().view(w.shape[0], -1)
        v = _pw_fwd(u, w, w2, f32_out)
        ctx.save_for_backward(u)
        ctx.wshape, ctx.param = w.shape, w
        return v

    @staticmethod
    def backward(ctx, dv):
        (u,) = ctx.saved_tensors
        w = ctx.param
        dv = _import(dv, u.dtype)                      # bf16 mode: the f32 logit gradient becomes a bf16 GEMM operand
        du, dw = _pw_bwd(dv, u, w, w.detach().to(torch.float32).contiguous().view(w.shape[0], -1))
        return du, None if dw is None else dw.view(ctx.wshape), None


class BatchNormTrain(torch.autograd.Function):
    """BatchNorm1d in train mode (+ optional ReLU): statistics over all B*T frames (quirk A4).  `running` = (running_mean,
    running_var, momentum, num_batches_tracked) or None: the module's running statistics, updated by the same launch."""

    @staticmethod
    def forward(ctx, v, gamma, beta, eps, relu, running):
        v = _import(v, v.dtype if is_act(v) else _ACT_DTYPE)
        g, be = gamma.detach().to(torch.float32).contiguous(), beta.detach().to(torch.float32).contiguous()
        b, c, t = v.shape
        y = alloc_like(v)
        mr = torch.empty(c, 2, dtype=torch.float32, device=v.device)
        ws = torch.empty(16 * c, dtype=torch.float64, device=v.device)
        rm, rv, mom, nbt = running if running is not None else (None, None, 0.0, None)
        st = _lib.lib().ts_train_bn_fwd(v.data_ptr(), g.data_ptr(), be.data_ptr(), y.data_ptr(), mr.data_ptr(), ws.data_ptr(), b, c, t, _pitch(v),
                                        float(eps), int(relu), rm.data_ptr() if rm is not None else None,
                                        rv.data_ptr() if rv is not None else None, float(mom),
                                        nbt.data_ptr() if nbt is not None else None, _code(v), _s(v))
        _lib.check(st, "ts_train_bn_fwd")
        ctx.save_for_backward(v, y, g, mr)
        ctx.relu, ctx.params = relu, (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        v, y, g, mr = ctx.saved_tensors
        dy = _g(dy, v)
        b, c, t = v.shape
        dv = alloc_like(v)
        dg, db = grad_out(ctx.params[0], (c,)), grad_out(ctx.params[1], (c,))
        ws = torch.empty(16 * c, dtype=torch.float64, device=v.device)
        st = _lib.lib().ts_train_bn_bwd(dy.data_ptr(), y.data_ptr(), v.data_ptr(), g.data_ptr(), mr.data_ptr(), dv.data_ptr(), dg.data_ptr(),
                                        db.data_ptr(), ws.data_ptr(), b, c, t, _pitch(v), int(ctx.relu), _code(v), _s(v))
        _lib.check(st, "ts_train_bn_bwd")
        return dv, dg, db, None, None, None


class AddRelu(torch.autograd.Function):
    """out = relu(a + b) (b may be None)."""

    @staticmethod
    def forward(ctx, a, b):
        a = _import(a, a.dtype if is_act(a) else _ACT_DTYPE)
        bb = _import(b, a.dtype) if b is not None else None
        out = alloc_like(a)
        st = _lib.lib().ts_train_add_relu_fwd(a.data_ptr(), bb.data_ptr() if bb is not None else None, out.data_ptr(), a.shape[0] * a.shape[1],
                                              a.shape[2], _pitch(a), _code(a), _s(a))
        _lib.check(st, "ts_train_add_relu_fwd")
        ctx.save_for_backward(out)
        ctx.has_b = b is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        dout = _g(dout, out)
        din = alloc_like(out)
        st = _lib.lib().ts_train_relu_bwd(dout.data_ptr(), out.data_ptr(), din.data_ptr(), out.shape[0] * out.shape[1], out.shape[2],
                                          _pitch(out), _code(out), _s(out))
        _lib.check(st, "ts_train_relu_bwd")
        return din, (din if ctx.has_b else None)


class Dropout(torch.autograd.Function):
    """nn.Dropout in train mode: y = x * keep / (1 - p), keep ~ Bernoulli(1 - p) drawn per element from a Philox stream
    (csrc/train_extra.hip).  Nothing is saved: the backward pass re-draws the mask from the same seed."""

    @staticmethod
    def forward(ctx, x, p, seed):
        x3 = x if x.dim() == 3 else x.reshape(1, 1, -1)
        x3 = _import(x3, x3.dtype if is_act(x3) else _ACT_DTYPE)
        y = alloc_like(x3)
        st = _lib.lib().ts_train_dropout(x3.data_ptr(), y.data_ptr(), x3.shape[0] * x3.shape[1], x3.shape[2], _pitch(x3), float(p), int(seed), _nonce(x3),
                                         _code(x3), _s(x3))
        _lib.check(st, "ts_train_dropout")
        ctx.p, ctx.seed, ctx.shape, ctx.dtype = float(p), int(seed), x.shape, x3.dtype
        return y if x.dim() == 3 else y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        d3 = dy if dy.dim() == 3 else dy.reshape(1, 1, -1)
        d3 = _import(d3, ctx.dtype)
        dx = alloc_like(d3)
        st = _lib.lib().ts_train_dropout(d3.data_ptr(), dx.data_ptr(), d3.shape[0] * d3.shape[1], d3.shape[2], _pitch(d3), ctx.p, ctx.seed, _nonce(d3),
                                         _code(d3), _s(d3))
        _lib.check(st, "ts_train_dropout")
        return (dx if len(ctx.shape) == 3 else dx.reshape(ctx.shape)), None, None


def dropout(x: Tensor, p: float, training: bool) -> Tensor:
    if not training or p <= 0.0:
        return x
    from .rng import next_seed
    return Dropout.apply(x, p, next_seed())


class SubsampleMask(torch.autograd.Function):
    """Input side of a strided 1x1 MaskedConv1d: zero the frames >= length, keep every `stride`-th frame."""

    @staticmethod
    def forward(ctx, x, lens, stride, t_out):
        x = _import(x, x.dtype if is_act(x) else _ACT_DTYPE)
        b, c, t_in = x.shape
        y = alloc(b, c, t_out, x.device, x.dtype)
        st = _lib.lib().ts_train_subsample_mask(x.data_ptr(), lens.data_ptr(), y.data_ptr(), b, c, t_in, t_out, stride, 0, _pitch(x), _pitch(y),
                                                _code(x), _s(x))
        _lib.check(st, "ts_train_subsample_mask")
        ctx.save_for_backward(lens)
        ctx.geom = (b, c, t_in, t_out, stride, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        (lens,) = ctx.saved_tensors
        b, c, t_in, t_out, stride, dtype = ctx.geom
        dy = _import(dy, dtype)
        dx = alloc(b, c, t_in, dy.device, dtype)
        st = _lib.lib().ts_train_subsample_mask(dy.data_ptr(), lens.data_ptr(), dx.data_ptr(), b, c, t_in, t_out, stride, 1, _pitch(dx), _pitch(dy),
                                                _code(dy), _s(dy))
        _lib.check(st, "ts_train_subsample_mask")
        return dx, None, None, None


class SqueezeExciteTrain(torch.autograd.Function):
    """SqueezeExcite.forward with autograd (citrinet/blocks.py:70-83): y = x * sigmoid(W2 relu(W1 mean_t(x))), the mean over ALL
    frames (quirk A3).  Every step is a HIP launch (csrc/train_extra.hip): the passes over the activation and the [B, C] bottleneck
    with its backward (ts_train_se_gate_fwd / _bwd); torch only allocates."""

    @staticmethod
    def forward(ctx, x, w1, w2):
        x = _import(x, x.dtype if is_act(x) else _ACT_DTYPE)
        # any floating dtype / stride is accepted, as nn.Linear's own matmul would (the kernels take contiguous f32: a copy when needed; the
        # gradients come back in f32 and autograd casts them to the parameters' dtype).  A bottleneck too wide for the gate kernels' LDS
        # (C + hidden > 16 K floats; Citrinet-1024 has 1 152) raises NotImplementedError from _lib.check -- loudly, there is no ATen path.
        w1, w2 = w1.detach().to(torch.float32).contiguous(), w2.detach().to(torch.float32).contiguous()
        b, c, t = x.shape
        r = w1.shape[0]
        if tuple(w1.shape) != (r, c) or tuple(w2.shape) != (c, r):
            raise ValueError(f"SqueezeExciteTrain: weights {tuple(w1.shape)}, {tuple(w2.shape)} do not fit {c} channels")
        L = _lib.lib()
        mean = torch.empty(b, c, dtype=torch.float32, device=x.device)
        h = torch.empty(b, r, dtype=torch.float32, device=x.device)
        g = torch.empty(b, c, dtype=torch.float32, device=x.device)
        _lib.check(L.ts_train_se_pool(x.data_ptr(), mean.data_ptr(), b * c, t, _pitch(x), _code(x), _s(x)), "ts_train_se_pool")
        _lib.check(L.ts_train_se_gate_fwd(mean.data_ptr(), w1.data_ptr(), w2.data_ptr(), h.data_ptr(), g.data_ptr(), b, c, r, _s(x)), "ts_train_se_gate_fwd")
        y = alloc_like(x)
        _lib.check(L.ts_train_se_scale(x.data_ptr(), g.data_ptr(), None, y.data_ptr(), b * c, t, _pitch(x), _code(x), _s(x)), "ts_train_se_scale")
        ctx.save_for_backward(x, w1, w2, mean, h, g)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, mean, h, g = ctx.saved_tensors
        dy = _g(dy, x)
        b, c, t = x.shape
        r = w1.shape[0]
        L = _lib.lib()
        f32 = dict(dtype=torch.float32, device=x.device)
        dg, dz, dmean = torch.empty(b, c, **f32), torch.empty(b, c, **f32), torch.empty(b, c, **f32)
        dh, dw1, dw2 = torch.empty(b, r, **f32), torch.empty(r, c, **f32), torch.empty(c, r, **f32)
        _lib.check(L.ts_train_se_rowdot(dy.data_ptr(), x.data_ptr(), dg.data_ptr(), b * c, t, _pitch(x), _code(x), _s(x)), "ts_train_se_rowdot")
        _lib.check(L.ts_train_se_gate_bwd(dg.data_ptr(), g.data_ptr(), h.data_ptr(), mean.data_ptr(), w1.data_ptr(), w2.data_ptr(), dz.data_ptr(),
                                          dh.data_ptr(), dmean.data_ptr(), dw1.data_ptr(), dw2.data_ptr(), b, c, r, _s(x)), "ts_train_se_gate_bwd")
        dx = alloc_like(x)
        _lib.check(L.ts_train_se_scale(dy.data_ptr(), g.data_ptr(), dmean.data_ptr(), dx.data_ptr(), b * c, t, _pitch(x), _code(x), _s(x)),
                   "ts_train_se_scale")
        return dx, dw1, dw2


class SubBlockCfg:
    """Non-tensor arguments of SubBlock (one per call)."""
    __slots__ = ("len_in", "len_out", "k", "stride", "dil", "pad", "eps", "relu", "running", "drop_p", "drop_seed",
                 "lazy_in", "lazy_out", "out_sums", "bwd_mask", "tile_stats", "defer_stats")


# BatchNorm between two repeats folded into the neighbouring depthwise launches (no normalised tensor in memory): on by default
_LAZY_BN = True


def set_lazy_batchnorm(on: bool) -> None:
    global _LAZY_BN
    _LAZY_BN = bool(on)


def same_depthwise(conv) -> bool:
    """The geometry the pair kernels (and with them the folded BatchNorm) cover: MaskedConv1d depthwise, stride 1, dilation 1, odd K, same padding."""
    return (conv is not None and conv.stride == 1 and conv.dilation == 1 and conv.kernel_size % 2 == 1 and conv.kernel_size <= 128
            and conv.padding == (conv.kernel_size - 1) // 2 and conv.conv.in_channels % 2 == 0)


class SubBlock(torch.autograd.Function):
    """One repeat of a QuartzNet / Citrinet block as ONE autograd node (quartznet/blocks.py:195-228): [depthwise MaskedConv1d |
    mask] -> 1x1 MaskedConv1d -> BatchNorm1d(train) [-> ReLU] [-> Dropout].  What is fused is the host side (one node instead of
    four) and, between two repeats of a block, the BatchNorm itself:
      * `cfg.lazy_out`: the node stops after the clip-group SUMS of its 1x1 output v (ts_train_bn_stats) and returns v; mean / rstd,
        the running-statistics update and the normalisation + ReLU happen inside the next repeat's depthwise launch while it stages
        its input (`cfg.lazy_in` = (sums, relu, eps, running buffers) of that BatchNorm: ts_train_dwconv_fwd_bn) -- the normalised
        tensor never exists in memory;
      * backward: the next repeat's depthwise backward (ts_train_dwconv_bwd_bn) also forms the two sums of that BatchNorm's backward
        (= its dgamma / dbeta) and ts_train_bn_bwd_sums finishes it, so the node returns dL/dv complete."""

    @staticmethod
    def forward(ctx, x, dw_w, pw_w, gamma, beta, gamma_in, beta_in, cfg):
        L = _lib.lib()
        x = _import(x, x.dtype if is_act(x) else _ACT_DTYPE)
        b, c_in, t_in = x.shape
        st_, code = _s(x), _code(x)
        lazy = cfg.lazy_in
        if dw_w is not None:
            w_dw = dw_w.detach().to(torch.float32).contiguous().view(dw_w.shape[0], -1)
            t_out = (t_in + 2 * cfg.pad - cfg.dil * (cfg.k - 1) - 1) // cfg.stride + 1
            mid = alloc(b, c_in, t_out, x.device, x.dtype)
            if lazy is not None:
                g_in, b_in = gamma_in.detach().to(torch.float32).contiguous(), beta_in.detach().to(torch.float32).contiguous()
                in_mr = torch.empty(c_in, 2, dtype=torch.float32, device=x.device)
                sums, relu_in, eps_in, run_in = lazy
                rm_i, rv_i, mom_i, nbt_i = run_in if run_in is not None else (None, None, 0.0, None)
                if sums.dtype == torch.float32:      # per-tile pairs out of the producing 1x1 launch's epilogue (tile_stats_buffer)
                    st_t = L.ts_train_dwconv_fwd_bn_tiles(x.data_ptr(), sums.data_ptr(), sums.shape[1], g_in.data_ptr(), b_in.data_ptr(), float(eps_in), int(relu_in),
                                                          in_mr.data_ptr(), rm_i.data_ptr() if rm_i is not None else None,
                                                          rv_i.data_ptr() if rv_i is not None else None, float(mom_i),
                                                          nbt_i.data_ptr() if nbt_i is not None else None,
                                                          cfg.len_in.data_ptr(), cfg.len_out.data_ptr(), w_dw.data_ptr(), mid.data_ptr(), b, c_in, t_in,
                                                          cfg.k, cfg.pad, _pitch(x), code, st_)
                    if st_t == _lib.TS_EUNSUPPORTED:  # a consumer without the tile reduction: fold the pairs into the clip-group layout (group 0 = all)
                        folded = torch.zeros(8, c_in, 2, dtype=torch.float64, device=x.device)
                        folded[0] = sums.to(torch.float64).sum(1)
                        sums = folded
                    else:
                        _lib.check(st_t, "ts_train_dwconv_fwd_bn_tiles")
                if sums.dtype != torch.float32:
                  _lib.check(L.ts_train_dwconv_fwd_bn(x.data_ptr(), sums.data_ptr(), g_in.data_ptr(), b_in.data_ptr(), float(eps_in), int(relu_in),
                                                    in_mr.data_ptr(), rm_i.data_ptr() if rm_i is not None else None,
                                                    rv_i.data_ptr() if rv_i is not None else None, float(mom_i),
                                                    nbt_i.data_ptr() if nbt_i is not None else None,
                                                    cfg.len_in.data_ptr(), cfg.len_out.data_ptr(), w_dw.data_ptr(), mid.data_ptr(), b, c_in, t_in,
                                                    cfg.k, cfg.pad, _pitch(x), code, st_), "ts_train_dwconv_fwd_bn")
            else:
                g_in = b_in = in_mr = None
                _lib.check(L.ts_train_dwconv_fwd(x.data_ptr(), cfg.len_in.data_ptr(), cfg.len_out.data_ptr(), w_dw.data_ptr(), mid.data_ptr(), b, c_in,
                                                 t_in, t_out, cfg.k, cfg.stride, cfg.dil, cfg.pad, _pitch(x), _pitch(mid), code, st_), "ts_train_dwconv_fwd")
        else:
            if lazy is not None:
                raise RuntimeError("SubBlock: a folded BatchNorm input needs a depthwise convolution")
            w_dw, t_out, g_in, b_in, in_mr = None, t_in, None, None, None
            if _pw_masks_inside(x.dtype, c_in, pw_w.shape[0]):
                mid = x                  # the 1x1 kernels (forward and weight gradient) mask by length themselves: no masked copy
            else:
                mid = alloc_like(x)
                _lib.check(L.ts_train_mask_time(x.data_ptr(), cfg.len_in.data_ptr(), mid.data_ptr(), b, c_in, t_in, _pitch(x), _pitch(mid), code, st_),
                           "ts_train_mask_time")
        w_pw = pw_w.detach().to(torch.float32).contiguous().view(pw_w.shape[0], -1)
        c_out = w_pw.shape[0]
        inside = dw_w is None and mid is x
        # statistics for a BatchNorm the NEXT repeat's matrix-core depthwise launch will fold in: per-tile pairs out of this launch's epilogue
        # instead of a pass of their own over v (93 launches of 5.5 us per QuartzNet15x5 step at 32 x 501 frames)
        tiles = None
        if cfg.lazy_out and getattr(cfg, "tile_stats", False) and TILE_STATS and mid.dtype == torch.bfloat16 and b >= 17 and _tcs_ok(True, c_in):
            tiles = tile_stats_buffer(b, c_out, t_out, x.device)
            if tiles.shape[1] > 512:     # every consumer wave sums all the tiles of its channel: past ~512 the separate pass is cheaper
                tiles = None             # (local batch 256 x 501 frames = 2 048 tiles: 54.2 vs 52.8 ms per step, measured)
        v = _pw_fwd(mid, pw_w, w_pw, lens=cfg.len_in if inside else None, stats=tiles)
        ws = torch.empty(16 * c_out, dtype=torch.float64, device=x.device)
        g, be = gamma.detach().to(torch.float32).contiguous(), beta.detach().to(torch.float32).contiguous()
        mr = torch.empty(c_out, 2, dtype=torch.float32, device=x.device)
        rm, rv, mom, nbt = cfg.running if cfg.running is not None else (None, None, 0.0, None)
        if cfg.lazy_out and tiles is not None:
            cfg.out_sums = tiles
            y, out = None, v
        elif cfg.lazy_out and getattr(cfg, "defer_stats", False) and chan_fits(b, t_out, v.dtype):
            cfg.out_sums = None      # the block tail's one-launch kernel forms the statistics itself (ts_train_bn2_add_relu_chan_fwd)
            y, out = None, v
        elif cfg.lazy_out:
            _lib.check(L.ts_train_bn_stats(v.data_ptr(), ws.data_ptr(), b, c_out, t_out, _pitch(v), code, st_), "ts_train_bn_stats")
            cfg.out_sums = ws
            y, out = None, v
        else:
            y = alloc_like(v)
            _lib.check(L.ts_train_bn_fwd(v.data_ptr(), g.data_ptr(), be.data_ptr(), y.data_ptr(), mr.data_ptr(), ws.data_ptr(), b, c_out, t_out, _pitch(v),
                                         float(cfg.eps), int(cfg.relu), rm.data_ptr() if rm is not None else None,
                                         rv.data_ptr() if rv is not None else None, float(mom), nbt.data_ptr() if nbt is not None else None, code, st_),
                       "ts_train_bn_fwd")
            out = y
            if cfg.drop_p > 0.0:
                out = alloc_like(y)
                _lib.check(L.ts_train_dropout(y.data_ptr(), out.data_ptr(), b * c_out, t_out, _pitch(y), float(cfg.drop_p), int(cfg.drop_seed), _nonce(x), code, st_),
                           "ts_train_dropout")
        saved = [x, mid, v, g, mr] + ([y] if y is not None else []) + ([w_dw] if w_dw is not None else []) + ([in_mr, g_in, b_in] if lazy is not None else [])
        ctx.save_for_backward(*saved)
        ctx.cfg, ctx.params = cfg, (dw_w, pw_w, gamma, beta, gamma_in, beta_in)
        ctx.shapes = (None if dw_w is None else dw_w.shape, pw_w.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        saved = list(ctx.saved_tensors)
        cfg = ctx.cfg
        x, mid, v, g, mr = saved[:5]
        rest = saved[5:]
        y = rest.pop(0) if not cfg.lazy_out else None
        dw_p, pw_p, ga_p, be_p, gin_p, bin_p = ctx.params
        w_dw = rest.pop(0) if dw_p is not None else None
        in_mr, g_in, b_in = rest if cfg.lazy_in is not None else (None, None, None)
        b, c_in, t_in = x.shape
        c_out, t_out = v.shape[1], v.shape[2]
        st_, code = _s(x), _code(x)
        if cfg.lazy_out:
            dv, dg, db = _g(dy, v), None, None          # the next repeat already took this BatchNorm's backward: dy IS dL/dv
        else:
            dy = _g(dy, y)
            if cfg.drop_p > 0.0:
                d2 = alloc_like(dy)
                _lib.check(L.ts_train_dropout(dy.data_ptr(), d2.data_ptr(), b * c_out, t_out, _pitch(dy), float(cfg.drop_p), int(cfg.drop_seed), _nonce(x), code, st_),
                           "ts_train_dropout")
                dy = d2
            dv = alloc_like(v)
            dg, db = grad_out(ga_p, (c_out,)), grad_out(be_p, (c_out,))
            ws = torch.empty(16 * c_out, dtype=torch.float64, device=x.device)
            _lib.check(L.ts_train_bn_bwd(dy.data_ptr(), y.data_ptr(), v.data_ptr(), g.data_ptr(), mr.data_ptr(), dv.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                         ws.data_ptr(), b, c_out, t_out, _pitch(v), int(cfg.relu), code, st_), "ts_train_bn_bwd")
        inside = dw_p is None and mid.data_ptr() == x.data_ptr()          # forward fed the unmasked x to kernels that mask inside
        dmid, dpw = _pw_bwd(dv, mid, pw_p, pw_p.detach().to(torch.float32).contiguous().view(pw_p.shape[0], -1), len_u=cfg.len_in if inside else None)
        dx = alloc_like(x)
        dg_in = db_in = None
        if w_dw is not None:
            # a frozen depthwise weight (callbacks.FinetuneEncoderDecoder's first phase) takes the data-gradient-only form of the kernel
            ddw = grad_out(dw_p, w_dw.shape, zeroed=True) if dw_p.requires_grad else None
            ddw_ptr = ddw.data_ptr() if ddw is not None else None
            if cfg.lazy_in is not None:
                gbuf = alloc_like(x)
                dg_in, db_in = grad_out(gin_p, (c_in,), zeroed=True), grad_out(bin_p, (c_in,), zeroed=True)
                _lib.check(L.ts_train_dwconv_bwd_bn(dmid.data_ptr(), x.data_ptr(), in_mr.data_ptr(), g_in.data_ptr(), b_in.data_ptr(), int(cfg.lazy_in[1]),
                                                    cfg.len_in.data_ptr(), cfg.len_out.data_ptr(), w_dw.data_ptr(), gbuf.data_ptr(), ddw_ptr,
                                                    dg_in.data_ptr(), db_in.data_ptr(), b, c_in, t_in, cfg.k, cfg.pad, _pitch(x), code, st_),
                           "ts_train_dwconv_bwd_bn")
                _lib.check(L.ts_train_bn_bwd_sums(gbuf.data_ptr(), x.data_ptr(), g_in.data_ptr(), in_mr.data_ptr(), dg_in.data_ptr(), db_in.data_ptr(),
                                                  dx.data_ptr(), b, c_in, t_in, _pitch(x), code, st_), "ts_train_bn_bwd_sums")
            else:
                # the block input needs no gradient (the stem: its input are the features) and the geometry is one whose data gradient is a
                # launch of its own (stride > 1): skip it (55 us of a QuartzNet15x5 step at 32 x 1001 x 64)
                skip_dx = not ctx.needs_input_grad[0] and cfg.stride != 1 and ddw_ptr is not None
                _lib.check(L.ts_train_dwconv_bwd(dmid.data_ptr(), x.data_ptr(), cfg.len_in.data_ptr(), cfg.len_out.data_ptr(), w_dw.data_ptr(),
                                                 None if skip_dx else dx.data_ptr(), ddw_ptr, b, c_in, t_in, t_out, cfg.k, cfg.stride, cfg.dil, cfg.pad,
                                                 _pitch(x), _pitch(dmid), code, st_), "ts_train_dwconv_bwd")
                if skip_dx:
                    dx = None
            ddw = ddw.view(ctx.shapes[0]) if ddw is not None else None
        else:
            ddw = None
            if cfg.bwd_mask:
                _lib.check(L.ts_train_mask_time(dmid.data_ptr(), cfg.len_in.data_ptr(), dx.data_ptr(), b, c_in, t_in, _pitch(dmid), _pitch(dx), code, st_),
                           "ts_train_mask_time")
            else:
                dx = dmid            # the consumer of this gradient (Fork with res_len) applies the input mask's backward
        return dx, ddw, None if dpw is None else dpw.view(ctx.shapes[1]), dg, db, dg_in, db_in, None


TILE_STATS = True           # False: every pending BatchNorm gets its statistics from ts_train_bn_stats (a pass over the tensor), as before ABI v8 (A/B)


def sub_block(x: Tensor, dw_conv, pw_conv, bn: torch.nn.BatchNorm1d, len_in: Tensor, len_out: Tensor, relu: bool, drop_p: float = 0.0,
              lazy_out: bool = False, bwd_mask: bool = True, tile_stats: bool = False, defer_stats: bool = False) -> Tensor:
    """x -> [dropout](relu?(BN_train(pw(mask(dw(mask(x))))))): one repeat of a block.  dw_conv / pw_conv are the MaskedConv1d modules
    (dw_conv None for a non-separable 1x1 repeat), len_in / len_out int32 device lengths before / after the depthwise conv.
    `lazy_out` (only between two repeats, see SubBlock): the result is the UN-normalised 1x1 output carrying its pending BatchNorm
    (`._ts_lazy`); hand it to the next sub_block call and to nothing else.  `tile_stats` (with lazy_out, when the consumer is the next repeat's
    depthwise launch and not a block tail): the statistics may travel as the per-tile pairs of the 1x1 launch (ts_tcs_desc.stats)."""
    cfg = SubBlockCfg()
    cfg.len_in, cfg.len_out, cfg.bwd_mask = len_in, len_out, bool(bwd_mask)   # bwd_mask False: x comes out of Fork(x, res_len), which masks the gradient
    if dw_conv is not None:
        cfg.k, cfg.stride, cfg.dil, cfg.pad = dw_conv.kernel_size, dw_conv.stride, dw_conv.dilation, dw_conv.padding
    else:
        cfg.k, cfg.stride, cfg.dil, cfg.pad = 1, 1, 1, 0
    cfg.eps, cfg.relu, cfg.running = bn.eps, relu, _running(bn)
    cfg.drop_p, cfg.drop_seed = (float(drop_p), 0)
    if cfg.drop_p > 0.0:
        from .rng import next_seed
        cfg.drop_seed = next_seed()
    pending = getattr(x, "_ts_lazy", None)              # (sums, relu, bn module, running buffers) of the previous repeat's BatchNorm
    cfg.lazy_in = (pending[0], pending[1], pending[2].eps, pending[3]) if pending is not None else None
    cfg.lazy_out, cfg.out_sums = bool(lazy_out) and cfg.drop_p == 0.0, None
    cfg.tile_stats = bool(tile_stats)
    cfg.defer_stats = bool(defer_stats)          # `lazy_out` towards block_tail: leave the statistics to its one-launch kernel when the batch fits
    if pending is not None and not same_depthwise(dw_conv):
        raise RuntimeError("sub_block: the input carries a pending BatchNorm but this repeat cannot apply it")
    gamma_in, beta_in = (pending[2].weight, pending[2].bias) if pending is not None else (None, None)
    y = SubBlock.apply(x, None if dw_conv is None else dw_conv.conv.weight, pw_conv.conv.weight, bn.weight, bn.bias, gamma_in, beta_in, cfg)
    if pending is not None:
        _bump_running(pending[2], pending[3])           # this launch applied the previous BatchNorm's running-statistics update
    if cfg.lazy_out:
        y._ts_lazy = (cfg.out_sums, bool(relu), bn, cfg.running)
    else:
        _bump_running(bn, cfg.running)
    return y


CHAN_TAIL = True            # False: block tails keep the two-step kernels (clip-group sums + apply passes), as before ABI v9 (A/B)


def chan_fits(batch: int, t: int, dtype) -> bool:
    """Whether the one-workgroup-per-channel block-tail kernels hold a channel of this batch in registers (csrc/train_enc.hip ChanRegs)."""
    return CHAN_TAIL and batch * ((t + 511) // 512) <= (32 if dtype == torch.bfloat16 else 16)


class BlockTail(torch.autograd.Function):
    """out = relu(BatchNorm(v_main) + BatchNorm(v_res)) (quartznet/blocks.py:332-337) over the two un-normalised tensors (both branches end
    in a `lazy_out` sub_block).  When neither branch brought statistics (`defer_stats`, the batch fits): ONE launch forward
    (ts_train_bn2_add_relu_chan_fwd: statistics + both normalisations + add + ReLU, every tensor read once) and ONE launch backward
    (ts_train_bn2_chan_bwd: both BatchNorm backwards with the shared ReLU's gate taken from `out`); otherwise one apply pass over the
    clip-group sums forward and two (sums + apply) pairs backward."""

    @staticmethod
    def forward(ctx, va, gamma_a, beta_a, vb, gamma_b, beta_b, cfg):
        L = _lib.lib()
        b, c, t = va.shape
        out = alloc_like(va)
        ga, ba = gamma_a.detach().to(torch.float32).contiguous(), beta_a.detach().to(torch.float32).contiguous()
        gb, bb = gamma_b.detach().to(torch.float32).contiguous(), beta_b.detach().to(torch.float32).contiguous()
        mra, mrb = (torch.empty(c, 2, dtype=torch.float32, device=va.device) for _ in range(2))
        chan = cfg[0][0] is None and cfg[1][0] is None and chan_fits(b, t, va.dtype)
        args = []
        for v, sums, g, be, eps, mr, run in ((va, cfg[0][0], ga, ba, cfg[0][1], mra, cfg[0][2]), (vb, cfg[1][0], gb, bb, cfg[1][1], mrb, cfg[1][2])):
            rm, rv, mom, nbt = run if run is not None else (None, None, 0.0, None)
            if not chan and sums is None:         # one branch came without statistics: the pass it skipped
                sums = torch.empty(16 * c, dtype=torch.float64, device=v.device)
                _lib.check(L.ts_train_bn_stats(v.data_ptr(), sums.data_ptr(), b, c, t, _pitch(v), _code(v), _s(v)), "ts_train_bn_stats")
            args += [v.data_ptr()] + ([] if chan else [sums.data_ptr()]) + [g.data_ptr(), be.data_ptr(), float(eps), mr.data_ptr(),
                     rm.data_ptr() if rm is not None else None, rv.data_ptr() if rv is not None else None, float(mom), nbt.data_ptr() if nbt is not None else None]
        if chan:
            _lib.check(L.ts_train_bn2_add_relu_chan_fwd(*args, out.data_ptr(), b, c, t, _pitch(va), _code(va), _s(va)), "ts_train_bn2_add_relu_chan_fwd")
        else:
            _lib.check(L.ts_train_bn2_add_relu_fwd(*args, out.data_ptr(), b, c, t, _pitch(va), _code(va), _s(va)), "ts_train_bn2_add_relu_fwd")
        ctx.save_for_backward(va, vb, out, ga, gb, mra, mrb)
        ctx.params = (gamma_a, beta_a, gamma_b, beta_b)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        va, vb, out, ga, gb, mra, mrb = ctx.saved_tensors
        b, c, t = va.shape
        d2, len2 = take_pending_add(dout)
        dout = _g(dout, out)
        if d2 is not None and not chan_fits(b, t, va.dtype):      # cannot happen (Fork checks the same condition); kept correct anyway
            summed = alloc_like(dout)
            _lib.check(L.ts_train_add(dout.data_ptr(), d2.data_ptr(), len2.data_ptr() if len2 is not None else None, c, summed.data_ptr(), b * c, t,
                                      _pitch(dout), _code(dout), _s(dout)), "ts_train_add")
            dout, d2 = summed, None
        if chan_fits(b, t, va.dtype):
            dva, dvb = alloc_like(va), alloc_like(vb)
            dga, dba = grad_out(ctx.params[0], (c,)), grad_out(ctx.params[1], (c,))
            dgb, dbb = grad_out(ctx.params[2], (c,)), grad_out(ctx.params[3], (c,))
            _lib.check(L.ts_train_bn2_chan_bwd(dout.data_ptr(), d2.data_ptr() if d2 is not None else None, len2.data_ptr() if (d2 is not None and len2 is not None) else None,
                                               out.data_ptr(), va.data_ptr(), vb.data_ptr(), ga.data_ptr(), mra.data_ptr(), gb.data_ptr(),
                                               mrb.data_ptr(), dva.data_ptr(), dvb.data_ptr(), dga.data_ptr(), dba.data_ptr(), dgb.data_ptr(), dbb.data_ptr(),
                                               b, c, t, _pitch(va), _code(va), _s(va)), "ts_train_bn2_chan_bwd")
            return dva, dga, dba, dvb, dgb, dbb, None
        grads = []
        for v, g, mr, (gp, bp) in ((va, ga, mra, ctx.params[:2]), (vb, gb, mrb, ctx.params[2:])):
            dv = alloc_like(v)
            dg, db = grad_out(gp, (c,)), grad_out(bp, (c,))
            ws = torch.empty(16 * c, dtype=torch.float64, device=v.device)
            _lib.check(L.ts_train_bn_bwd(dout.data_ptr(), out.data_ptr(), v.data_ptr(), g.data_ptr(), mr.data_ptr(), dv.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                         ws.data_ptr(), b, c, t, _pitch(v), 1, _code(v), _s(v)), "ts_train_bn_bwd")
            grads += [dv, dg, db]
        return (*grads, None)


def block_tail(h: Tensor, r: Tensor) -> Tensor:
    """relu(BN(h) + BN(r)) for two `lazy_out` sub_block results (main branch, residual branch)."""
    ph, pr = getattr(h, "_ts_lazy", None), getattr(r, "_ts_lazy", None)
    if ph is None or pr is None or ph[1] or pr[1]:
        raise RuntimeError("block_tail: both inputs must carry a pending BatchNorm without ReLU")
    cfg = ((ph[0], ph[2].eps, ph[3]), (pr[0], pr[2].eps, pr[3]))
    out = BlockTail.apply(h, ph[2].weight, ph[2].bias, r, pr[2].weight, pr[2].bias, cfg)
    if cfg[0][0] is None and cfg[1][0] is None and chan_fits(h.shape[0], h.shape[2], h.dtype):
        out._ts_tail_out = True          # a Fork that takes this tensor may leave the sum of its two gradients to this tail's backward kernel
    _bump_running(ph[2], ph[3])
    _bump_running(pr[2], pr[3])
    return out


def _running(bn: torch.nn.BatchNorm1d):
    if not (bn.track_running_stats and bn.running_mean is not None):
        return None
    if bn.running_mean.dtype != torch.float32 or bn.running_var.dtype != torch.float32 or not bn.running_mean.is_cuda:
        raise RuntimeError("batch_norm_train: fp32 running statistics on the GPU only")
    # momentum=None (cumulative average) needs the counter's value: one host read, the reference default is 0.1
    m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
    return (bn.running_mean, bn.running_var, m, bn.num_batches_tracked)


def _bump_running(bn, running) -> None:
    if running is not None:
        # the launch updated the buffers through raw pointers: make the change visible to `_version`-keyed caches
        # (blocks._PackedCache folds running_mean / running_var into the inference weights)
        for t in (bn.running_mean, bn.running_var, bn.num_batches_tracked):
            torch.autograd.graph.increment_version(t)


def batch_norm_train(bn: torch.nn.BatchNorm1d, v: Tensor, relu: bool) -> Tensor:
    """BatchNorm1d(train) through the kernels + the module's running-statistics update (momentum, unbiased variance)."""
    running = _running(bn)
    y = BatchNormTrain.apply(v, bn.weight, bn.bias, bn.eps, relu, running)
    _bump_running(bn, running)
    return y
