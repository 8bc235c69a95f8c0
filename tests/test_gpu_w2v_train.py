"""GPU parity of wav2vec2 FINE-TUNING (thunder_speech_amd/huggingface/train.py, csrc/w2v_train.hip, csrc/gemm_f32.hip) against the REAL
`transformers.Wav2Vec2Model` in train mode with torch autograd on the CPU: the reference fine-tunes HuggingFace checkpoints through
BaseCTCModule.training_step with the conv feature extractor frozen (/root/reference/tests/huggingface/test_module_huggingface.py:33-54,
src/thunder/huggingface/compatibility.py:23-42).  Seeded random weights (pretrained ones need the network); transformers 5.15 in this image
(the reference pins 4.21: same modelling code for these layers)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
transformers = pytest.importorskip("transformers")

BASE = dict(vocab_size=32, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, conv_dim=(32, 32, 32),
            conv_stride=(5, 2, 2), conv_kernel=(10, 3, 2), num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, hidden_dropout=0.0,
            activation_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0, final_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0,
            mask_feature_prob=0.0)
FAMILIES = {"group_postln": dict(feat_extract_norm="group", do_stable_layer_norm=False, conv_bias=False),
            "layer_preln": dict(feat_extract_norm="layer", do_stable_layer_norm=True, conv_bias=True),
            # head_dim 64 and 64 channels per positional-conv group: the geometry on which mixed precision takes the fused attention (csrc/w2v_attn_train.hip) and the
            # matrix-core positional conv (ts_w2v_posconv_train / _wgrad) -- the whole model against transformers' autograd with those kernels in the path
            "group_postln_hd64": dict(feat_extract_norm="group", do_stable_layer_norm=False, conv_bias=False, hidden_size=128, num_attention_heads=2,
                                      intermediate_size=256, num_conv_pos_embedding_groups=2)}


def _model(family, seed=0, **over):
    torch.manual_seed(seed)
    cfg = transformers.Wav2Vec2Config(**{**BASE, **FAMILIES[family], **over})
    m = transformers.Wav2Vec2Model(cfg)
    with torch.no_grad():                       # default init leaves the positional conv and LayerNorms near-trivial: make every parameter matter
        for n, p in m.named_parameters():
            if p.dim() == 1 and "bias" in n:
                p.add_(0.05 * torch.randn_like(p))
            elif "layer_norm.weight" in n:
                p.mul_(1.0 + 0.1 * torch.randn_like(p))
        if hasattr(m, "masked_spec_embed"):           # created only with mask_time_prob / mask_feature_prob > 0
            m.masked_spec_embed.copy_(torch.randn_like(m.masked_spec_embed))
    return m


def _pair(family, seed=0, **over):
    """(reference model on the CPU, the same weights behind the HIP adapter on the GPU), both in train mode, feature extractor frozen."""
    from thunder_speech_amd.huggingface.encoder import HuggingFaceEncoderAdapt
    ref = _model(family, seed, **over)
    ref.freeze_feature_encoder()
    ref.train()
    mine = _model(family, seed, **over)
    mine.load_state_dict(ref.state_dict())
    adapt = HuggingFaceEncoderAdapt(mine, mask_input=over.get("mask_input", False), precision="fp32").cuda().train()
    return ref, adapt


def _inputs(b=3, n=4000, seed=1):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(b, n, generator=g)


def _compare_grads(ref, adapt, tol=3e-3):
    mine = dict(adapt.original_encoder.named_parameters())
    worst = 0.0
    checked = 0
    # k_proj.bias has NO true gradient (it shifts every score of a query by the same amount, which the softmax ignores): both sides hold rounding
    # noise there, so gradients are measured against the larger of their own norm and 1e-4 of the largest gradient norm of the model
    floor = 1e-4 * max(float(p.grad.norm()) for p in ref.parameters() if p.grad is not None)
    for name, p in ref.named_parameters():
        q = mine[name]
        if not p.requires_grad:
            assert q.grad is None or float(q.grad.abs().max()) == 0.0, name        # frozen feature extractor: no gradient
            continue
        assert p.grad is not None and q.grad is not None, name
        denom = max(float(p.grad.norm()), floor)
        rel = float((q.grad.cpu() - p.grad).norm()) / denom
        worst = max(worst, rel)
        assert rel <= tol, (name, rel)
        checked += 1
    assert checked > 20
    return worst


@pytest.mark.parametrize("family", list(FAMILIES))
@pytest.mark.parametrize("masked", [False, True])
def test_training_forward_and_every_gradient_match_transformers_autograd(family, masked):
    """All dropouts off: last_hidden_state (5e-4) and the gradient of EVERY trainable parameter (relative L2 3e-3) of a scalar loss, with and
    without an attention mask (ragged clips).  Covers LayerNorm / linear / attention / GELU / positional conv (weight-norm g and v) backward."""
    ref, adapt = _pair(family)
    adapt.mask_input = masked
    x = _inputs()
    lengths = torch.tensor([4000, 3000, 2111])
    if masked:
        x = x * (torch.arange(x.shape[1])[None, :] < lengths[:, None])
    att = (torch.arange(x.shape[1])[None, :] < lengths[:, None]).long() if masked else None
    out_ref = ref(x, attention_mask=att).last_hidden_state
    probe = torch.randn(out_ref.shape, generator=torch.Generator().manual_seed(5))
    (out_ref * probe).sum().backward()
    feats, out_len = adapt(x.cuda(), lengths.cuda())
    assert feats.shape == (3, out_ref.shape[2], out_ref.shape[1])
    got = feats.transpose(-1, -2)
    np.testing.assert_allclose(got.detach().cpu().numpy(), out_ref.detach().numpy(), atol=5e-4, rtol=1e-4)
    (got * probe.cuda()).sum().backward()
    _compare_grads(ref, adapt)


def test_time_masking_follows_transformers_under_the_same_numpy_seed():
    """mask_time_prob > 0 (Wav2Vec2Model._mask_hidden_states): the span geometry is transformers' `_compute_mask_indices` restated on numpy's
    global RNG, so the same seed masks the same frames -- outputs and gradients (incl. masked_spec_embed's) match."""
    ref, adapt = _pair("group_postln", mask_time_prob=0.3, mask_time_length=3, mask_time_min_masks=2)
    x = _inputs(b=2, n=6000, seed=2)
    np.random.seed(11)
    out_ref = ref(x).last_hidden_state
    probe = torch.randn(out_ref.shape, generator=torch.Generator().manual_seed(6))
    (out_ref * probe).sum().backward()
    np.random.seed(11)
    feats, _ = adapt(x.cuda(), torch.tensor([6000, 6000]).cuda())
    got = feats.transpose(-1, -2)
    np.testing.assert_allclose(got.detach().cpu().numpy(), out_ref.detach().numpy(), atol=5e-4, rtol=1e-4)
    (got * probe.cuda()).sum().backward()
    _compare_grads(ref, adapt)
    assert float(adapt.original_encoder.masked_spec_embed.grad.abs().max()) > 0


def test_layerdrop_skips_the_same_layers_as_transformers_under_the_same_torch_seed():
    ref, adapt = _pair("group_postln", layerdrop=0.5, num_hidden_layers=4)
    x = _inputs(b=2, n=3000, seed=3)
    torch.manual_seed(21)
    out_ref = ref(x).last_hidden_state
    torch.manual_seed(21)
    feats, _ = adapt(x.cuda(), torch.tensor([3000, 3000]).cuda())
    np.testing.assert_allclose(feats.transpose(-1, -2).detach().cpu().numpy(), out_ref.detach().numpy(), atol=5e-4, rtol=1e-4)


def test_dropouts_are_random_scaled_and_differentiable():
    """hidden / activation / attention / feature-projection dropout on the device's Philox stream: two forwards differ, the mean over many
    elements is preserved (1 / (1 - p) scaling), the backward runs and reaches every trainable parameter."""
    _, adapt = _pair("group_postln", hidden_dropout=0.1, activation_dropout=0.1, attention_dropout=0.1, feat_proj_dropout=0.1)
    x = _inputs(b=2, n=4000, seed=4).cuda()
    ln = torch.tensor([4000, 4000]).cuda()
    a, _ = adapt(x, ln)
    b, _ = adapt(x, ln)
    assert not torch.equal(a, b) and torch.isfinite(a).all()
    adapt.eval()
    with torch.no_grad():
        e, _ = adapt(x, ln)
    adapt.train()
    assert float((a.detach() - e).abs().mean()) < 0.5                           # LayerNorm-ed outputs of unit scale: dropout perturbs, it does not destroy
    a.square().mean().backward()
    for name, p in adapt.original_encoder.named_parameters():
        if p.requires_grad and "masked_spec_embed" not in name:
            assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0, name


def test_ctc_training_step_through_the_module_updates_the_transformer_and_not_the_feature_extractor():
    """The reference's fine-tuning loop in miniature (tests/huggingface/test_module_huggingface.py:33-54): BaseCTCModule.training_step on an
    HF encoder + linear decoder, three AdamW steps: the loss falls, the feature extractor's weights stay, the transformer's move."""
    from thunder_speech_amd.blocks import linear_decoder
    from thunder_speech_amd.huggingface.encoder import HuggingFaceEncoderAdapt
    from thunder_speech_amd.huggingface.transform import Wav2Vec2Preprocess
    from thunder_speech_amd.module import BaseCTCModule
    from thunder_speech_amd.text_processing.transform import BatchTextTransformer
    enc = HuggingFaceEncoderAdapt(_model("group_postln", seed=3), precision="fp32")
    tokens = [chr(97 + i) for i in range(26)] + [" "]
    module = BaseCTCModule(enc, linear_decoder(64, len(tokens) + 1, 0.0), Wav2Vec2Preprocess(), BatchTextTransformer(tokens=tokens),
                           optimizer_kwargs={"lr": 1e-3}).cuda().train()
    before = {n: p.detach().clone() for n, p in enc.original_encoder.named_parameters()}
    opt = module.configure_optimizers()
    batch = (_inputs(b=2, n=8000, seed=7).cuda(), torch.tensor([8000.0, 6000.0]).cuda(), ["hello world", "abc"])
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = module.training_step(batch, 0)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0] and all(np.isfinite(losses))
    moved = {n: float((p.detach() - before[n]).abs().max()) for n, p in enc.original_encoder.named_parameters()}
    assert all(v == 0.0 for n, v in moved.items() if n.startswith("feature_extractor."))
    assert moved["encoder.layers.1.attention.q_proj.weight"] > 0 and moved["feature_projection.projection.weight"] > 0


def test_unfrozen_feature_extractor_is_refused_and_the_frozen_plan_holds_only_its_weights():
    """The reference freezes the conv feature extractor once, in __init__ (compatibility.py:27-28).  Flipping requires_grad back on afterwards would
    make autograd train it in the reference; this path has no backward there and must say so rather than return no gradient."""
    from thunder_speech_amd.huggingface.encoder import HuggingFaceEncoderAdapt
    adapt = HuggingFaceEncoderAdapt(_model("group_postln", seed=5), precision="fp32").cuda().train()
    x = _inputs(b=2, n=6000, seed=1).cuda()
    adapt(x, torch.tensor([6000, 6000]).cuda())
    plan = adapt._plan_frozen(x.device)
    assert plan.feature_extractor_only and plan.layers == [] and not hasattr(plan, "fp_w")
    with pytest.raises(RuntimeError):
        plan.forward(x, None)
    next(adapt.original_encoder.feature_extractor.parameters()).requires_grad_(True)
    with pytest.raises(NotImplementedError, match="feature extractor is frozen"):
        adapt(x, torch.tensor([6000, 6000]).cuda())


@pytest.mark.parametrize("family", list(FAMILIES))
def test_mixed_precision_training_matches_transformers_autograd_within_bf16_tolerance(family):
    """train_precision="bf16" (huggingface/train.py LinearMixed: bf16 operands + f32 accumulation in the linear layers' forward, data-gradient and
    weight-gradient products on csrc/gemm_nt.hip; everything else f32) -- the reference under Lightning's precision="bf16-mixed".  Stated
    tolerance against f32 autograd through the real transformers model: last_hidden_state within 3e-2 of its unit scale, every parameter
    gradient within 4e-2 relative L2 (bf16 operands carry 2^-9 relative rounding through 2 layers x 6 products); the f32 mode's 3e-3 / 5e-4
    are checked above.  The ragged (attention-masked) form is the one run here; the forward is bit-reproducible."""
    ref, adapt = _pair(family)
    adapt.train_precision = "bf16"
    adapt.mask_input = True
    x = _inputs()
    lengths = torch.tensor([4000, 3000, 2111])
    x = x * (torch.arange(x.shape[1])[None, :] < lengths[:, None])
    att = (torch.arange(x.shape[1])[None, :] < lengths[:, None]).long()
    out_ref = ref(x, attention_mask=att).last_hidden_state
    probe = torch.randn(out_ref.shape, generator=torch.Generator().manual_seed(5))
    (out_ref * probe).sum().backward()
    feats, _ = adapt(x.cuda(), lengths.cuda())
    got = feats.transpose(-1, -2)
    err = float((got.detach().cpu() - out_ref.detach()).abs().max())
    assert err <= 3e-2 * max(1.0, float(out_ref.abs().max())), err
    assert err > 1e-6, "the mixed-precision path did not run (bit-level agreement with f32 is not what bf16 operands give)"
    if family.endswith("hd64"):                      # the fused kernels are in this graph
        from thunder_speech_amd.huggingface import train as T
        seen, todo = set(), [got.grad_fn]
        while todo:
            fn = todo.pop()
            if fn is None or fn in seen:
                continue
            seen.add(fn)
            todo += [f for f, _ in fn.next_functions]
        names = {type(fn).__name__ for fn in seen}
        assert {"AttentionFusedBackward", "FFNActLinearBackward", "PosConvGeluBackward"} <= names, names
    (got * probe.cuda()).sum().backward()
    worst = _compare_grads(ref, adapt, tol=4e-2)
    assert worst > 1e-5
    with torch.no_grad():
        adapt.train()
        feats2, _ = adapt(x.cuda(), lengths.cuda())
    assert torch.equal(feats2, feats)


def test_cast_bf16_t_writes_the_plain_and_the_zero_padded_transposed_copy():
    from thunder_speech_amd import _lib
    L = _lib.lib()
    for rows, c in ((597, 64), (70, 96), (1, 32), (130, 200)):
        x = torch.randn(rows, c, device="cuda")
        rp = (rows + 31) // 32 * 32
        y = torch.full((rows, c), 7.0, dtype=torch.bfloat16, device="cuda")
        yt = torch.full((c, rp), 7.0, dtype=torch.bfloat16, device="cuda")
        _lib.check(L.ts_w2v_cast_bf16_t(x.data_ptr(), c, rows, c, y.data_ptr(), c, yt.data_ptr(), rp, rp, torch.cuda.current_stream().cuda_stream), "cast")
        assert torch.equal(y, x.to(torch.bfloat16))
        assert torch.equal(yt[:, :rows], x.to(torch.bfloat16).t()) and float(yt[:, rows:].float().abs().sum()) == 0.0


def test_vectorised_dropout_draws_the_oracles_philox_stream():
    """huggingface/train.py Dropout on f32 [rows, c] tensors takes ts_train_dropout's four-elements-per-thread form (c % 4 == 0): one Philox block per
    four elements -- the same keep mask as the oracle's per-element definition (word e & 3 of block e >> 2, e = row * c + i), bit for bit; an odd width
    takes the scalar form and agrees with the oracle too."""
    from oracle import philox as ph
    from thunder_speech_amd.huggingface.train import Dropout
    for rows, c, p in ((37, 1024, 0.1), (5, 4096, 0.5), (9, 1023, 0.3)):
        x = torch.randn(rows, c, device="cuda") + 3.0
        y = Dropout.apply(x, p, 424242)
        keep = torch.from_numpy(ph.dropout_keep(424242, rows * c, p)).view(rows, c).cuda()
        assert torch.equal(y != 0, keep)
        torch.testing.assert_close(y[keep], x[keep] / (1.0 - p), rtol=1e-6, atol=0)


def test_split_k_product_equals_the_single_launch():
    from thunder_speech_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(3)
    for rows, n, k, splits in ((1024, 1024, 4096, 8), (512, 256, 2048, 4), (96, 64, 1024, 2)):
        a = torch.randn(rows, k, device="cuda", generator=g).to(torch.bfloat16)
        w = torch.randn(n, k, device="cuda", generator=g).to(torch.bfloat16)
        one = torch.empty(rows, n, device="cuda")
        _lib.check(L.ts_gemm_nt_bf16(a.data_ptr(), k, w.data_ptr(), k, None, None, 0, one.data_ptr(), n, None, 0, rows, n, k, 0, st), "gemm")
        parts = torch.empty(splits, rows, n, device="cuda")
        out = torch.empty(rows, n, device="cuda")
        _lib.check(L.ts_gemm_nt_bf16_splitk(a.data_ptr(), k, w.data_ptr(), k, parts.data_ptr(), rows, n, k, splits, st), "splitk")
        _lib.check(L.ts_w2v_sum_parts(parts.data_ptr(), out.data_ptr(), rows * n, splits, st), "sum")
        ref = a.float() @ w.float().t()
        scale = float(ref.abs().max())
        assert float((one - ref).abs().max()) <= 2e-3 * scale and float((out - ref).abs().max()) <= 2e-3 * scale
        assert L.ts_gemm_nt_bf16_splitk(a.data_ptr(), k, w.data_ptr(), k, parts.data_ptr(), rows, n, k, 3, st) != 0      # k / 3 is not a multiple of 32


@pytest.mark.parametrize("b,t,heads,p,ragged", [(2, 130, 4, 0.0, False), (3, 499, 4, 0.0, True), (2, 499, 4, 0.1, False), (3, 200, 2, 0.25, True), (1, 33, 1, 0.5, False)])
def test_fused_training_attention_matches_the_materialised_path_forward_and_backward(b, t, heads, p, ragged):
    """csrc/w2v_attn_train.hip (no [T][T] matrix in either direction; bf16 operands) against huggingface/train.py's Attention (f32 products over materialised
    probabilities): the context, the row statistic and dq / dk / dv within 2e-2 relative L2 -- WITH dropout too, i.e. the mask drawn inside the three fused
    kernels is ts_train_dropout's for the same seed (a different mask would show as an error of the order of p) -- for ragged key lengths incl. a clip
    with no valid key, frame counts that are no multiple of the 64-key / 128-query tiles, and every element of dqkv written."""
    from thunder_speech_amd.huggingface import train as T
    torch.manual_seed(b * 100 + t)
    c = 64 * heads
    qkv = (torch.randn(b, t, 3 * c, device="cuda") * 1.5).requires_grad_(True)
    key_len = torch.tensor([t, t // 2, 0][:b], dtype=torch.int32, device="cuda") if ragged else None
    seed = 987654321012345
    ref = T.Attention.apply(qkv, key_len, heads, p, seed)
    dout = torch.randn_like(ref)
    ref.backward(dout)
    dref = qkv.grad.clone()
    qkv.grad = None
    old = T._MIXED
    T._MIXED = True
    try:
        out = T.attention(qkv, key_len, heads, p, seed)
        assert isinstance(out.grad_fn, T.AttentionFused._backward_cls)
        out.backward(dout)
    finally:
        T._MIXED = old
    torch.cuda.synchronize()
    rel = lambda a_, r_: float((a_ - r_).norm() / r_.norm())
    assert rel(out.detach(), ref.detach()) <= 2e-2
    for sl in (slice(0, c), slice(c, 2 * c), slice(2 * c, 3 * c)):
        assert bool(torch.isfinite(qkv.grad[..., sl]).all())
        assert rel(qkv.grad[..., sl], dref[..., sl]) <= 2e-2
    if p > 0:                                            # the mask matters: another seed gives another result
        T._MIXED = True
        try:
            other = T.attention(qkv.detach(), key_len, heads, p, seed + 1)
        finally:
            T._MIXED = old
        assert rel(other, ref.detach()) > 5e-2


def test_fused_training_attention_is_reproducible_bit_for_bit():
    """No atomics in the fused attention kernels: two runs give the same bits (forward and all of dqkv)."""
    from thunder_speech_amd.huggingface import train as T
    torch.manual_seed(5)
    qkv = torch.randn(2, 300, 3 * 128, device="cuda")
    dout = torch.randn(2, 300, 128, device="cuda")
    res = []
    old = T._MIXED
    T._MIXED = True
    try:
        for _ in range(2):
            x = qkv.clone().requires_grad_(True)
            y = T.attention(x, None, 2, 0.1, 42)
            y.backward(dout)
            res.append((y.detach().clone(), x.grad.clone()))
    finally:
        T._MIXED = old
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("b,t", [(2, 130), (3, 499)])
def test_mixed_precision_positional_conv_matches_the_f32_path_forward_and_backward(b, t):
    """PosConvGelu in mixed mode (forward and data gradient on the bf16 matrix-core kernel: ts_w2v_posconv_train; the transposed conv as the same product over
    flipped, per-tap transposed weights and a copy of dz padded kernel - 1 - kernel / 2 rows in front) against its f32 products: y, dx, dw, db within 1.5e-2
    relative L2; the bias gradient rides in the cast launch of LinearMixed the same way (ts_w2v_cast_bf16_t_colsum: checked against the separate pass)."""
    from thunder_speech_amd import _lib
    from thunder_speech_amd.huggingface import train as T
    torch.manual_seed(t)
    c, k, g = 1024, 128, 16
    x = torch.randn(b, t, c, device="cuda")
    wk = (torch.randn(k, g, 64, 64, device="cuda") * (64 * k) ** -0.5)
    bias = torch.randn(c, device="cuda") * 0.1
    dy = torch.randn(b, t, c, device="cuda")
    res = {}
    old = T._MIXED
    try:
        for mode in (False, True):
            T._MIXED = mode
            xi, wi, bi = x.clone().requires_grad_(True), wk.clone().requires_grad_(True), bias.clone().requires_grad_(True)
            y = T.PosConvGelu.apply(xi, wi, bi)
            y.backward(dy)
            res[mode] = (y.detach(), xi.grad, wi.grad, bi.grad)
    finally:
        T._MIXED = old
    torch.cuda.synchronize()
    for a_, r_, what in zip(res[True], res[False], ("y", "dx", "dw", "db")):
        assert bool(torch.isfinite(a_).all()), what
        assert float((a_ - r_).norm() / r_.norm()) <= 1.5e-2, what
    # the column sums out of the cast launch
    L = _lib.lib()
    rows, n = 777, 320
    m = torch.randn(rows, n, device="cuda")
    out16 = torch.empty(rows, n, dtype=torch.bfloat16, device="cuda")
    cs = torch.zeros(n, device="cuda")
    assert L.ts_w2v_cast_bf16_t_colsum(m.data_ptr(), n, rows, n, out16.data_ptr(), n, None, 0, rows, cs.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(out16, m.to(torch.bfloat16))
    assert float((cs - m.sum(0)).abs().max()) <= 1e-3


def test_fused_attention_backward_gives_the_same_bits_with_the_kept_and_the_redrawn_mask():
    """ts_w2v_attention_train_bwd(fwd_mask = the forward's workspace) against fwd_mask = NULL (the mask re-drawn from the seed): the same dqkv, bit for bit."""
    import ctypes as C
    from thunder_speech_amd import _lib
    L = _lib.lib()
    torch.manual_seed(2)
    b, t, heads, p, seed = 2, 211, 3, 0.2, 77
    c = 64 * heads
    st = torch.cuda.current_stream().cuda_stream
    q16 = torch.randn(b, t, 3 * c, device="cuda").to(torch.bfloat16)
    dout = torch.randn(b, t, c, device="cuda")
    ctx_, lse2 = torch.empty(b, t, c, device="cuda"), torch.empty(b, heads, t, device="cuda")
    wsf = torch.empty(L.ts_w2v_attention_train_fwd_workspace(b, t, c, heads), dtype=torch.uint8, device="cuda")
    assert L.ts_w2v_attention_train_fwd(q16.data_ptr(), b, t, c, heads, None, p, seed, ctx_.data_ptr(), lse2.data_ptr(), wsf.data_ptr(), st) == 0
    outs = []
    for mask in (wsf, None):
        ws = torch.empty(L.ts_w2v_attention_train_bwd_workspace(b, t, c, heads), dtype=torch.uint8, device="cuda")
        dqkv = torch.full((b, t, 3 * c), float("nan"), device="cuda")
        assert L.ts_w2v_attention_train_bwd(q16.data_ptr(), b, t, c, heads, None, p, seed, dout.data_ptr(), ctx_.data_ptr(), lse2.data_ptr(),
                                            mask.data_ptr() if mask is not None else None, dqkv.data_ptr(), ws.data_ptr(), st) == 0
        torch.cuda.synchronize()
        outs.append(dqkv)
    assert torch.equal(outs[0], outs[1]) and bool(torch.isfinite(outs[0]).all())


@pytest.mark.parametrize("rows_shape,k,n,p", [((2, 130), 256, 128, 0.1), ((3, 499), 4096, 1024, 0.1), ((1, 77), 512, 96, 0.0)])
def test_fused_feed_forward_activation_node_equals_the_three_separate_nodes(rows_shape, k, n, p):
    """FFNActLinear (gelu + dropout + the second linear's operand cast in one launch, ts_w2v_ffn_act_cast; dropout-backward x gelu' in one launch,
    ts_w2v_ffn_act_bwd) against BiasGelu -> Dropout -> LinearMixed with the same seed: the same bf16 operands, so the same output and the same gradients
    (bit for bit for y, dz, dw; the bias gradients are atomic sums: 1e-5)."""
    from thunder_speech_amd.huggingface import train as T
    torch.manual_seed(k + n)
    z = torch.randn(*rows_shape, k, device="cuda")
    b1 = torch.randn(k, device="cuda") * 0.1
    w2 = torch.randn(n, k, device="cuda") * k ** -0.5
    b2 = torch.randn(n, device="cuda") * 0.1
    dy = torch.randn(*rows_shape, n, device="cuda")
    seed = 424242
    res = []
    old = T._MIXED
    T._MIXED = True
    try:
        for fused in (True, False):
            zi, b1i, w2i, b2i = (t_.clone().requires_grad_(True) for t_ in (z, b1, w2, b2))
            if fused:
                y = T.FFNActLinear.apply(zi, b1i, w2i, b2i, p, seed)
            else:
                a = T.BiasGelu.apply(zi, b1i)
                a = T.Dropout.apply(a, p, seed) if p > 0 else a
                y = T.LinearMixed.apply(a, w2i, b2i)
            y.backward(dy)
            res.append((y.detach(), zi.grad, w2i.grad, b1i.grad, b2i.grad))
    finally:
        T._MIXED = old
    torch.cuda.synchronize()
    for i in (0, 1, 2):
        assert torch.equal(res[0][i], res[1][i]), i
    for i in (3, 4):
        assert torch.allclose(res[0][i], res[1][i], rtol=1e-5, atol=1e-5 * float(res[1][i].abs().max()))
    assert float(res[0][1].abs().sum()) > 0


@pytest.mark.parametrize("b,t,c,k,g", [(1, 37, 128, 16, 2), (2, 130, 256, 13, 4), (3, 257, 1024, 128, 16)])
def test_positional_conv_matrix_core_entry_points_against_f64_on_odd_geometries(b, t, c, k, g):
    """ts_w2v_posconv_train (forward incl. the pre-activation z; data gradient as the same product over flipped, transposed taps) and ts_w2v_posconv_wgrad
    against float64 einsums of the bf16-rounded operands: one clip, frame counts below / not multiples of the 128-frame tile, an ODD kernel (padding k / 2 on
    both sides, nothing dropped) and one that is no multiple of the 8 taps a workgroup owns."""
    from thunder_speech_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(b * 1000 + t)
    cg = c // g
    assert cg == 64
    x = torch.randn(b, t, c, device="cuda")
    wk = torch.randn(k, g, cg, cg, device="cuda") * (cg * k) ** -0.5
    bias = torch.randn(c, device="cuda") * 0.1
    dz = torch.randn(b, t, c, device="cuda")
    bf = lambda v: v.to(torch.bfloat16).double()
    # reference: y_conv[b][t][g cg + o] = sum_j sum_i x[b][t + j - k // 2][g cg + i] wk[j][g][o][i]
    xp = torch.zeros(b, t + k, c, dtype=torch.float64, device="cuda")
    xp[:, k // 2: k // 2 + t] = bf(x)
    win = torch.stack([xp[:, j: j + t] for j in range(k)], 0).view(k, b, t, g, cg)                      # [j][b][t][g][i]
    conv = torch.einsum("jbtgi,jgoi->btgo", win, bf(wk)).reshape(b, t, c)
    ws = torch.empty(L.ts_w2v_posconv_train_workspace(b, t, c, k), dtype=torch.uint8, device="cuda")
    y, z = torch.empty_like(x), torch.empty_like(x)
    assert L.ts_w2v_posconv_train(x.data_ptr(), x.data_ptr(), b, t, c, wk.to(torch.bfloat16).data_ptr(), bias.data_ptr(), k, g, 0, y.data_ptr(), z.data_ptr(), ws.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert float((z.double() - conv).abs().max()) <= 2e-5 * max(1.0, float(conv.abs().max()))
    y_ref = x.double() + torch.nn.functional.gelu(conv + bias.double())
    assert float((y.double() - y_ref).abs().max()) <= 1e-4
    # data gradient: dx[b][s][g cg + i] = dy + sum_j sum_o dz[b][s - j + k // 2][g cg + o] wk[j][g][o][i]
    dzp = torch.zeros(b, t + k, c, dtype=torch.float64, device="cuda")
    front = k - 1 - k // 2
    dzp[:, front: front + t] = bf(dz)
    wb = wk.flip(0).transpose(2, 3).contiguous()
    winz = torch.stack([dzp[:, j: j + t] for j in range(k)], 0).view(k, b, t, g, cg)
    dconv = torch.einsum("jbtgo,jgio->btgi", winz, bf(wb)).reshape(b, t, c)
    dy = torch.randn(b, t, c, device="cuda")
    dx = torch.empty_like(x)
    assert L.ts_w2v_posconv_train(dz.data_ptr(), dy.data_ptr(), b, t, c, wb.to(torch.bfloat16).data_ptr(), None, k, g, 1, dx.data_ptr(), None, ws.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert float((dx.double() - (dy.double() + dconv)).abs().max()) <= 2e-5 * max(1.0, float(dconv.abs().max()))
    # the definition of the transposed product, independently: autograd of the forward reference
    xr = bf(x).requires_grad_(True)
    xpr = torch.nn.functional.pad(xr, (0, 0, k // 2, k - k // 2))
    winr = torch.stack([xpr[:, j: j + t] for j in range(k)], 0).view(k, b, t, g, cg)
    torch.einsum("jbtgi,jgoi->btgo", winr, bf(wk)).reshape(b, t, c).backward(bf(dz))
    assert float((dconv - xr.grad).abs().max()) <= 1e-9 * max(1.0, float(xr.grad.abs().max()))
    # weight gradient
    dw = torch.full_like(wk, float("nan"))
    ws2 = torch.empty(L.ts_w2v_posconv_wgrad_workspace(b, t, c, k), dtype=torch.uint8, device="cuda")
    assert L.ts_w2v_posconv_wgrad(dz.data_ptr(), x.data_ptr(), b, t, c, k, g, dw.data_ptr(), ws2.data_ptr(), st) == 0
    torch.cuda.synchronize()
    dw_ref = torch.einsum("btgo,jbtgi->jgoi", bf(dz).view(b, t, g, cg), win)
    assert bool(torch.isfinite(dw).all())
    assert float((dw.double() - dw_ref).abs().max()) <= 2e-5 * max(1.0, float(dw_ref.abs().max()))


def test_layernorm_backward_set_form_equals_the_accumulating_form():
    """ts_w2v_layernorm_bwd_set (dgamma / dbeta zeroed by the first launch, one reducing launch) against ts_w2v_layernorm_bwd on zero-filled outputs; the
    outputs handed to the set form hold NaN before the call."""
    from thunder_speech_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(4)
    rows, c = 3992, 1024
    x, res, dy = (torch.randn(rows, c, device="cuda") for _ in range(3))
    gamma = torch.rand(c, device="cuda") + 0.5
    ws = torch.empty(L.ts_w2v_layernorm_bwd_workspace(rows, c), dtype=torch.uint8, device="cuda")
    out = []
    for fn, init in ((L.ts_w2v_layernorm_bwd, 0.0), (L.ts_w2v_layernorm_bwd_set, float("nan"))):
        dx = torch.empty_like(x)
        dg, db = torch.full((c,), init, device="cuda"), torch.full((c,), init, device="cuda")
        assert fn(x.data_ptr(), res.data_ptr(), gamma.data_ptr(), dy.data_ptr(), 1e-5, rows, c, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), st) == 0
        torch.cuda.synchronize()
        out.append((dx, dg, db))
    assert torch.equal(out[0][0], out[1][0])
    for i in (1, 2):
        assert torch.allclose(out[0][i], out[1][i], rtol=1e-5, atol=1e-4)
