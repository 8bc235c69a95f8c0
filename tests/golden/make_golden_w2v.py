"""Generates tests/golden/w2v_tiny.npz / w2v_tiny_layer.npz / w2v_mid.npz / hubert_tiny.npz / d2v_tiny.npz from the REAL transformers Wav2Vec2Model, HubertModel and Data2VecAudioModel (run in the
build container only): small configurations with seeded random weights, one unmasked and one masked forward pass each.
w2v_mid has head_dim 64 and 64 channels per positional-conv group, the geometry at which the HIP path takes its fused
attention / MFMA positional-conv kernels (the tiny ones exercise the fallback kernels).  Its 1 M-element positional-conv
direction tensor is not stored: it is drawn from a seeded torch generator here and re-drawn by tests/test_oracle_w2v.load_fixture
(same torch build on both sides; a stored checksum guards against generator drift)."""
import numpy as np
import torch
from transformers import Data2VecAudioConfig, Data2VecAudioModel, HubertConfig, HubertModel, Wav2Vec2Config, Wav2Vec2Model

CFG = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, feat_extract_norm="group",
           do_stable_layer_norm=False, vocab_size=32, conv_dim=(32,) * 7, conv_kernel=(10, 3, 3, 3, 3, 2, 2),
           conv_stride=(5, 2, 2, 2, 2, 2, 2), num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)


LAYER = dict(CFG, feat_extract_norm="layer", do_stable_layer_norm=True, conv_bias=True)
MID = dict(CFG, hidden_size=128, num_attention_heads=2, intermediate_size=256, num_conv_pos_embeddings=128,
           num_conv_pos_embedding_groups=2)
# HubertModel without the feature projection's LayerNorm (hubert-base); Data2VecAudioModel with its stacked positional convs
HUBERT = dict(CFG, feat_proj_layer_norm=False, model_type="hubert")
D2V = dict({k: v for k, v in CFG.items() if k not in ("feat_extract_norm", "do_stable_layer_norm")}, num_conv_pos_embeddings=5,
           conv_pos_kernel_size=19, model_type="data2vec-audio")
# Wav2Vec2Adapter behind the encoder: projection + LayerNorm to 48 channels, two strided conv + GLU layers
ADAPTER = dict(CFG, add_adapter=True, num_adapter_layers=2, adapter_kernel_size=3, adapter_stride=2, output_hidden_size=48)
MODELS = {"wav2vec2": (Wav2Vec2Config, Wav2Vec2Model), "hubert": (HubertConfig, HubertModel), "data2vec-audio": (Data2VecAudioConfig, Data2VecAudioModel)}
REGEN_SEED, REGEN_SCALE = 4321, 0.05


def main(cfg=CFG, name="w2v_tiny.npz", regen=()):
    torch.manual_seed(0)
    config_cls, model_cls = MODELS[cfg.get("model_type", "wav2vec2")]
    model = model_cls(config_cls(**{k: v for k, v in cfg.items() if k != "model_type"})).eval()
    with torch.no_grad():                      # make the norms / biases non-trivial
        for k, v in model.state_dict().items():
            if k.endswith("layer_norm.weight"):
                v.copy_(1.0 + 0.2 * torch.randn_like(v))
            elif k.endswith(".bias"):
                v.copy_(0.1 * torch.randn_like(v))
            if any(k.endswith(r) for r in regen):
                v.copy_(REGEN_SCALE * torch.randn(v.shape, generator=torch.Generator().manual_seed(REGEN_SEED)))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 8000, generator=g)
    lengths = torch.tensor([8000, 5000])
    mask = (torch.arange(8000)[None, :] < lengths[:, None]).int()
    with torch.no_grad():
        out = model(x).last_hidden_state
        xm = x * mask
        out_masked = model(xm, attention_mask=mask).last_hidden_state
        feat = model.feature_extractor(x).transpose(1, 2)
    arrays = {"sd/" + k: v.numpy() for k, v in model.state_dict().items() if not any(k.endswith(r) for r in regen)}
    for k, v in model.state_dict().items():
        if any(k.endswith(r) for r in regen):
            arrays["regen/" + k] = np.asarray([REGEN_SEED] + list(v.shape), dtype=np.int64)
            arrays["regen_sum/" + k] = np.asarray(float(v.double().abs().sum()))
    arrays.update(x=x.numpy(), lengths=lengths.numpy(), out=out.numpy(), out_masked=out_masked.numpy(), feat=feat.numpy(),
                  out_lengths=model._get_feat_extract_output_lengths(lengths).numpy())
    for k, v in cfg.items():
        if isinstance(v, (int, tuple, bool)):
            arrays["cfg/" + k] = np.asarray(v)
        elif isinstance(v, str):
            arrays["cfgs/" + k] = np.asarray(v)
    np.savez_compressed(__file__.replace("make_golden_w2v.py", name), **arrays)


if __name__ == "__main__":
    main()
    main(LAYER, "w2v_tiny_layer.npz")
    main(MID, "w2v_mid.npz", regen=("pos_conv_embed.conv.parametrizations.weight.original1", "pos_conv_embed.conv.weight_v"))
    main(HUBERT, "hubert_tiny.npz")
    main(D2V, "d2v_tiny.npz")
    main(ADAPTER, "w2v_tiny_adapter.npz")
