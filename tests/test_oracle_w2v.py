"""The wav2vec2 oracle against the fixture produced by the real transformers model (tests/golden/make_golden_w2v.py)."""
import os

import numpy as np
import torch

from oracle import w2v as ow

HERE = os.path.dirname(os.path.abspath(__file__))


# group-norm / post-LN and layer-norm / stable-LN families at toy size (fallback kernels on the GPU), and a mid-size group-norm
# configuration with head_dim 64 and 64 channels per positional-conv group (fused attention / MFMA positional conv on the GPU)
# + HubertModel without the feature projection's LayerNorm, Data2VecAudioModel (five stacked positional convs)
# + Wav2Vec2Model with its adapter (projection + LayerNorm + two strided conv / GLU layers behind the encoder)
FIXTURES = ["w2v_tiny.npz", "w2v_tiny_layer.npz", "w2v_mid.npz", "hubert_tiny.npz", "d2v_tiny.npz", "w2v_tiny_adapter.npz"]


def load_fixture(name="w2v_tiny.npz"):
    z = np.load(os.path.join(HERE, "golden", name))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    for k in z.files:                                     # large tensors drawn from a seeded generator instead of being stored
        if k.startswith("regen/"):
            seed, *shape = [int(v) for v in z[k]]
            t = 0.05 * torch.randn(shape, generator=torch.Generator().manual_seed(seed))
            assert abs(float(t.double().abs().sum()) - float(z["regen_sum/" + k[6:]])) < 1e-6 * t.numel(), "torch generator drift"
            sd[k[6:]] = t
    c = {k[4:]: z[k] for k in z.files if k.startswith("cfg/")}
    cfg = ow.W2VConfig(conv_dim=tuple(int(v) for v in c["conv_dim"]), conv_kernel=tuple(int(v) for v in c["conv_kernel"]),
                       conv_stride=tuple(int(v) for v in c["conv_stride"]), hidden_size=int(c["hidden_size"]),
                       num_hidden_layers=int(c["num_hidden_layers"]), num_attention_heads=int(c["num_attention_heads"]),
                       intermediate_size=int(c["intermediate_size"]), num_conv_pos_embeddings=int(c["num_conv_pos_embeddings"]),
                       num_conv_pos_embedding_groups=int(c["num_conv_pos_embedding_groups"]),
                       feat_extract_norm=str(z["cfgs/feat_extract_norm"]) if "cfgs/feat_extract_norm" in z.files else "group",
                       do_stable_layer_norm=bool(c.get("do_stable_layer_norm", False)), conv_bias=bool(c.get("conv_bias", False)),
                       model_type=str(z["cfgs/model_type"]) if "cfgs/model_type" in z.files else "wav2vec2",
                       feat_proj_layer_norm=bool(c.get("feat_proj_layer_norm", True)), conv_pos_kernel_size=int(c.get("conv_pos_kernel_size", 19)),
                       add_adapter=bool(c.get("add_adapter", False)), adapter_kernel_size=int(c.get("adapter_kernel_size", 3)),
                       adapter_stride=int(c.get("adapter_stride", 2)), num_adapter_layers=int(c.get("num_adapter_layers", 3)),
                       output_hidden_size=int(c["output_hidden_size"]) if "output_hidden_size" in c else None)
    if cfg.model_type == "data2vec-audio":
        cfg.feat_extract_norm = "layer"                  # Data2VecAudioConvLayer: always conv -> LayerNorm -> GELU
    return z, sd, cfg


import pytest


@pytest.mark.parametrize("name", FIXTURES)
def test_feature_extractor_matches_transformers(name):
    z, sd, cfg = load_fixture(name)
    feat = ow.feature_extractor(cfg, sd, torch.from_numpy(z["x"]))
    np.testing.assert_allclose(feat.numpy(), z["feat"], atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("name", FIXTURES)
def test_forward_matches_transformers_unmasked(name):
    z, sd, cfg = load_fixture(name)
    out, key_len = ow.forward(cfg, sd, torch.from_numpy(z["x"]))
    assert key_len is None
    np.testing.assert_allclose(out.numpy(), z["out"], atol=5e-5, rtol=1e-5)


@pytest.mark.parametrize("name", FIXTURES)
def test_forward_matches_transformers_masked_and_lengths(name):
    z, sd, cfg = load_fixture(name)
    x, lengths = torch.from_numpy(z["x"]), torch.from_numpy(z["lengths"])
    xm = x * (torch.arange(x.shape[1])[None, :] < lengths[:, None])
    out, key_len = ow.forward(cfg, sd, xm, lengths)
    np.testing.assert_array_equal(ow.output_lengths(cfg, lengths, cfg.add_adapter).numpy(), z["out_lengths"])
    assert cfg.add_adapter or np.array_equal(key_len.numpy(), z["out_lengths"])
    np.testing.assert_allclose(out.numpy(), z["out_masked"], atol=5e-5, rtol=1e-5)


@pytest.mark.parametrize("family", ["unispeech", "unispeech-sat"])
@pytest.mark.parametrize("style", ["group", "layer"])
def test_unispeech_models_run_the_wav2vec2_arithmetic(family, style):
    """UniSpeechModel / UniSpeechSatModel (other AutoModelForCTC families the reference's loader accepts, huggingface/compatibility.py:77) against the
    wav2vec2 restatement on THEIR state dict: same keys, same forward pass -- which is why the HIP adapter takes them on the wav2vec2 plan."""
    transformers = pytest.importorskip("transformers")
    z, _, base = load_fixture("w2v_tiny.npz")
    kw = dict(hidden_size=base.hidden_size, num_hidden_layers=base.num_hidden_layers, num_attention_heads=base.num_attention_heads,
              intermediate_size=base.intermediate_size, vocab_size=32, conv_dim=tuple(base.conv_dim), conv_kernel=tuple(base.conv_kernel),
              conv_stride=tuple(base.conv_stride), num_conv_pos_embeddings=base.num_conv_pos_embeddings,
              num_conv_pos_embedding_groups=base.num_conv_pos_embedding_groups, feat_extract_norm=style, do_stable_layer_norm=style == "layer",
              conv_bias=style == "layer")
    config_cls, model_cls = {"unispeech": (transformers.UniSpeechConfig, transformers.UniSpeechModel),
                             "unispeech-sat": (transformers.UniSpeechSatConfig, transformers.UniSpeechSatModel)}[family]
    torch.manual_seed(0)
    model = model_cls(config_cls(**kw)).eval()
    assert model.config.model_type == family
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith(".bias"):
                v.copy_(0.1 * torch.randn_like(v))
    x, lengths = torch.from_numpy(z["x"]), torch.from_numpy(z["lengths"])
    mask = (torch.arange(x.shape[1])[None, :] < lengths[:, None]).int()
    with torch.no_grad():
        want = model(x).last_hidden_state
        want_masked = model(x * mask, attention_mask=mask).last_hidden_state
    cfg = ow.W2VConfig(**{**base.__dict__, "feat_extract_norm": style, "do_stable_layer_norm": style == "layer", "conv_bias": style == "layer",
                          "layer_norm_eps": float(model.config.layer_norm_eps)})
    sd = {k: v for k, v in model.state_dict().items()}
    got, _ = ow.forward(cfg, sd, x)
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=5e-5, rtol=1e-5)
    got_m, key_len = ow.forward(cfg, sd, x * mask, lengths)
    np.testing.assert_array_equal(key_len.numpy(), z["out_lengths"])
    np.testing.assert_allclose(got_m.numpy(), want_masked.numpy(), atol=5e-5, rtol=1e-5)
