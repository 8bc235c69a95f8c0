"""Generates tests/golden/w2v_tiny.npz from the REAL transformers Wav2Vec2Model (run in the build container only):
a tiny group-norm / post-LN configuration with seeded random weights, one unmasked and one masked forward pass."""
import numpy as np
import torch
from transformers import Wav2Vec2Config, Wav2Vec2Model

CFG = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, feat_extract_norm="group",
           do_stable_layer_norm=False, vocab_size=32, conv_dim=(32,) * 7, conv_kernel=(10, 3, 3, 3, 3, 2, 2),
           conv_stride=(5, 2, 2, 2, 2, 2, 2), num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)


LAYER = dict(CFG, feat_extract_norm="layer", do_stable_layer_norm=True, conv_bias=True)


def main(cfg=CFG, name="w2v_tiny.npz"):
    torch.manual_seed(0)
    model = Wav2Vec2Model(Wav2Vec2Config(**cfg)).eval()
    with torch.no_grad():                      # make the norms / biases non-trivial
        for k, v in model.state_dict().items():
            if k.endswith("layer_norm.weight"):
                v.copy_(1.0 + 0.2 * torch.randn_like(v))
            elif k.endswith(".bias"):
                v.copy_(0.1 * torch.randn_like(v))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 8000, generator=g)
    lengths = torch.tensor([8000, 5000])
    mask = (torch.arange(8000)[None, :] < lengths[:, None]).int()
    with torch.no_grad():
        out = model(x).last_hidden_state
        xm = x * mask
        out_masked = model(xm, attention_mask=mask).last_hidden_state
        feat = model.feature_extractor(x).transpose(1, 2)
    arrays = {"sd/" + k: v.numpy() for k, v in model.state_dict().items()}
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
