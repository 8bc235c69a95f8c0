"""GPU parity: wav2vec2 encoder (csrc/w2v_enc.hip through the C ABI) vs the oracle and the transformers-generated fixture.
fp32 activations, fp32 GEMMs on the library's own matrix-core kernel (csrc/gemm_f32.hip): tolerances are fp32 summation-order noise."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import w2v as ow
from test_oracle_w2v import FIXTURES, load_fixture

pytestmark = pytest.mark.gpu


def _hf_cfg(cfg: ow.W2VConfig):
    return SimpleNamespace(conv_dim=cfg.conv_dim, conv_kernel=cfg.conv_kernel, conv_stride=cfg.conv_stride,
                           hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                           num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                           num_conv_pos_embeddings=cfg.num_conv_pos_embeddings,
                           num_conv_pos_embedding_groups=cfg.num_conv_pos_embedding_groups, layer_norm_eps=cfg.layer_norm_eps,
                           feat_extract_norm=cfg.feat_extract_norm, do_stable_layer_norm=cfg.do_stable_layer_norm,
                           conv_bias=cfg.conv_bias, hidden_act="gelu", feat_extract_activation="gelu", model_type=cfg.model_type,
                           feat_proj_layer_norm=cfg.feat_proj_layer_norm, conv_pos_kernel_size=cfg.conv_pos_kernel_size,
                           add_adapter=cfg.add_adapter, adapter_kernel_size=cfg.adapter_kernel_size, adapter_stride=cfg.adapter_stride,
                           num_adapter_layers=cfg.num_adapter_layers, output_hidden_size=cfg.output_hidden_size)


def _plan(precision="fp32", name="w2v_tiny.npz"):
    from thunder_speech_amd.huggingface.encoder import Wav2Vec2Plan
    z, sd, cfg = load_fixture(name)
    return z, sd, cfg, Wav2Vec2Plan(_hf_cfg(cfg), sd, "cuda", precision=precision)


@pytest.mark.parametrize("name", FIXTURES)
def test_bf16_operand_mode_stays_within_bf16_tolerance_of_the_fp32_reference(name):
    """precision="bf16": GEMM operands rounded to bf16 (8 mantissa bits), fp32 accumulation and normalisations.  The
    outputs are LayerNorm-ed (unit scale): a few 1e-2 absolute (max 0.1, rms 0.01; the k = 128 positional conv of the mid-size fixture sums 8192 bf16 products per output) is what operand rounding through 7 conv
    layers and 2 transformer layers gives."""
    z, sd, cfg, plan = _plan("bf16", name)
    x, lengths = torch.from_numpy(z["x"]), torch.from_numpy(z["lengths"])
    out = plan.forward(x.cuda(), None).cpu().numpy()
    # data2vec-audio: five more bf16 convolutions, each re-normalised to unit scale by its LayerNorm, sit in front of the encoder: rms 0.011 measured
    rms_tol = 0.015 if cfg.model_type == "data2vec-audio" else 0.01
    assert np.abs(out - z["out"]).max() <= 0.1 and np.sqrt(np.mean((out - z["out"]) ** 2)) <= rms_tol
    xm = x * (torch.arange(x.shape[1])[None, :] < lengths[:, None])
    outm = plan.forward(xm.cuda(), lengths.cuda()).cpu().numpy()
    want = z["out_masked"]
    if cfg.model_type == "data2vec-audio":
        # frames beyond a clip's length: the stacked convs see zeros there and their affine-free LayerNorms blow the bf16 rounding of a nearly
        # constant row up to unit scale -- ill-conditioned in the reference's own arithmetic too; the valid frames carry the claim
        valid = np.arange(want.shape[1])[None, :] < z["out_lengths"][:, None]
        outm, want = outm[valid], want[valid]
    assert np.abs(outm - want).max() <= 0.1 and np.sqrt(np.mean((outm - want) ** 2)) <= rms_tol


@pytest.mark.parametrize("name", FIXTURES)
def test_feature_extractor_matches_transformers_fixture(name):
    z, sd, cfg, plan = _plan(name=name)
    feat = plan.feature_extractor(torch.from_numpy(z["x"]).cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(feat.cpu().numpy(), z["feat"], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("name", FIXTURES)
def test_forward_matches_transformers_fixture_unmasked(name):
    z, sd, cfg, plan = _plan(name=name)
    out = plan.forward(torch.from_numpy(z["x"]).cuda(), None)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), z["out"], atol=5e-4, rtol=1e-4)


@pytest.mark.parametrize("name", FIXTURES)
def test_forward_matches_transformers_fixture_masked(name):
    z, sd, cfg, plan = _plan(name=name)
    x, lengths = torch.from_numpy(z["x"]), torch.from_numpy(z["lengths"])
    xm = x * (torch.arange(x.shape[1])[None, :] < lengths[:, None])
    out = plan.forward(xm.cuda(), lengths.cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), z["out_masked"], atol=5e-4, rtol=1e-4)


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("n,lens", [(4000, [4000, 3999, 1500]), (16000, [16000, 9000])])
def test_other_lengths_match_the_oracle(n, lens, name):
    """Ragged clip lengths and frame counts that are not multiples of anything."""
    z, sd, cfg, plan = _plan(name=name)
    g = torch.Generator().manual_seed(n)
    x = torch.randn(len(lens), n, generator=g)
    lengths = torch.tensor(lens)
    xm = x * (torch.arange(n)[None, :] < lengths[:, None])
    ref, key_len = ow.forward(cfg, sd, xm, lengths)
    out = plan.forward(xm.cuda(), lengths.cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=5e-4, rtol=1e-4)


def test_adapter_has_the_reference_interface():
    """`_HuggingFaceEncoderAdapt` contract (huggingface/compatibility.py:23-42) on a real transformers module."""
    transformers = pytest.importorskip("transformers")
    from thunder_speech_amd.huggingface.encoder import HuggingFaceEncoderAdapt
    z, sd, cfg = load_fixture()
    hf_cfg = transformers.Wav2Vec2Config(
        hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
        intermediate_size=cfg.intermediate_size, feat_extract_norm="group", do_stable_layer_norm=False, vocab_size=32,
        conv_dim=tuple(cfg.conv_dim), conv_kernel=tuple(cfg.conv_kernel), conv_stride=tuple(cfg.conv_stride),
        num_conv_pos_embeddings=cfg.num_conv_pos_embeddings, num_conv_pos_embedding_groups=cfg.num_conv_pos_embedding_groups)
    model = transformers.Wav2Vec2Model(hf_cfg).eval()
    model.load_state_dict(sd)
    enc = HuggingFaceEncoderAdapt(model.cuda(), mask_input=False, precision="fp32").eval()
    assert "original_encoder.encoder.layers.0.attention.q_proj.weight" in enc.state_dict()
    x = torch.from_numpy(z["x"]).cuda()
    lengths = torch.from_numpy(z["lengths"]).cuda()
    with torch.no_grad():
        out, out_len = enc(x, lengths)
    assert out.shape == (2, cfg.hidden_size, z["out"].shape[1])
    np.testing.assert_array_equal(out_len.cpu().numpy(), z["out_lengths"])
    np.testing.assert_allclose(out.transpose(-1, -2).cpu().numpy(), z["out"], atol=5e-4, rtol=1e-4)


@pytest.mark.parametrize("b,t,heads,lens", [(2, 200, 2, None), (3, 333, 3, [333, 100, 1]), (1, 999, 4, None), (2, 64, 1, [64, 33])])
def test_fused_attention_matches_softmax_qk_v(b, t, heads, lens):
    """ts_w2v_attention_fwd, precision 1, head_dim 64 (the fused MFMA kernel): against fp32 attention on the same bf16 inputs."""
    from thunder_speech_amd import _lib
    L = _lib.lib()
    c = 64 * heads
    g = torch.Generator().manual_seed(t)
    qkv = (torch.randn(b, t, 3 * c, generator=g) * 1.5).to(torch.bfloat16)
    qf = qkv.float()
    q, k, v = [z.view(b, t, heads, 64).transpose(1, 2) for z in qf.split(c, dim=-1)]
    scores = (q @ k.transpose(-1, -2)) / 8.0
    key_len = None
    if lens is not None:
        key_len = torch.tensor(lens, dtype=torch.int32)
        pad = torch.arange(t)[None, :] >= key_len[:, None]
        scores = scores.masked_fill(pad[:, None, None, :], float("-inf"))
    ref = (torch.softmax(scores, -1) @ v).transpose(1, 2).reshape(b, t, c)
    dq = qkv.cuda()
    ctx = torch.zeros(b, t, c, dtype=torch.bfloat16, device="cuda")
    ws = torch.empty(L.ts_w2v_attention_workspace_bytes(b, t, heads, 1), dtype=torch.uint8, device="cuda")
    kl = key_len.cuda() if key_len is not None else None
    st = L.ts_w2v_attention_fwd(dq.data_ptr(), b, t, c, heads, kl.data_ptr() if kl is not None else None, 1, ctx.data_ptr(),
                                ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert st == 0
    torch.cuda.synchronize()
    got = ctx.float().cpu()
    # bf16 probabilities and a bf16 result: 2-3 significant digits
    assert float((got - ref).abs().max()) <= 0.03 * max(1.0, float(ref.abs().max()))
    assert float((got - ref).pow(2).mean().sqrt()) <= 0.006 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("b,t,groups,k", [(2, 300, 2, 128), (1, 999, 3, 128), (3, 77, 1, 16), (2, 130, 2, 7)])
def test_posconv_mfma_kernel_matches_conv1d(b, t, groups, k):
    """ts_w2v_posconv_fwd, precision 1, 64 channels per group (the implicit-GEMM MFMA kernel) against F.conv1d on the same
    bf16-rounded operands: y = x + gelu(conv(x) + bias), 'same' padding k // 2, last frame of an even kernel dropped."""
    import torch.nn.functional as F
    from thunder_speech_amd import _lib
    L = _lib.lib()
    c, cg = 64 * groups, 64
    g = torch.Generator().manual_seed(t + k)
    x = torch.randn(b, t, c, generator=g)
    w = torch.randn(c, cg, k, generator=g) * (1.0 / (cg * k) ** 0.5)
    bias = torch.randn(c, generator=g) * 0.1
    xr, wr = x.to(torch.bfloat16).float(), w.to(torch.bfloat16).float()
    conv = F.conv1d(xr.transpose(1, 2), wr, bias, padding=k // 2, groups=groups)[:, :, :t]
    ref = x + F.gelu(conv).transpose(1, 2)
    w_taps = w.view(groups, cg, cg, k).permute(3, 0, 1, 2).contiguous().to(torch.bfloat16).cuda()
    xd, bd = x.cuda(), bias.cuda()
    y = torch.empty_like(xd)
    ws = torch.empty(L.ts_w2v_posconv_workspace_bytes(b, t, c, k), dtype=torch.uint8, device="cuda")
    st = L.ts_w2v_posconv_fwd(xd.data_ptr(), b, t, c, w_taps.data_ptr(), bd.data_ptr(), k, groups, 1, y.data_ptr(), None, ws.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
    assert st == 0
    torch.cuda.synchronize()
    assert float((y.cpu() - ref).abs().max()) <= 2e-3


@pytest.mark.parametrize("act", [0, 1])
def test_fused_epilogue_linear_matches_unfused_path(act):
    """ts_w2v_linear_fwd with bf16 operands and only the bf16 result wanted (library GEMM with the bias / bias+GELU epilogue
    fused) against exact fp32 arithmetic on the same bf16 operands: the result is bf16, so agreement is to ~1 bf16 ulp --
    that also bounds what the library's GELU may differ from the erf form by."""
    import torch.nn.functional as F
    from thunder_speech_amd import _lib
    L = _lib.lib()
    rows, k, n = 999 * 2, 256, 512
    g = torch.Generator().manual_seed(act)
    x = torch.randn(rows, k, generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, generator=g) * 0.5
    ref = x.float() @ w.float().T + b
    if act:
        ref = F.gelu(ref)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    y = torch.empty(rows, n, dtype=torch.float32, device="cuda")
    y16 = torch.zeros(rows, n, dtype=torch.bfloat16, device="cuda")
    st = L.ts_w2v_linear_fwd(xd.data_ptr(), k, wd.data_ptr(), bd.data_ptr(), None, n, y.data_ptr(), n, y16.data_ptr(), rows, n, k, act | 2, 1,
                             None, torch.cuda.current_stream().cuda_stream)
    assert st == 0
    torch.cuda.synchronize()
    err = (y16.float().cpu() - ref).abs()
    assert float(err.max()) <= 0.02 * max(1.0, float(ref.abs().max())) and float(err.mean()) <= 4e-3
