"""load_huggingface_checkpoint on a checkpoint directory written offline by transformers' own save_pretrained (random
weights): module structure / vocabulary on the CPU, logits and transcripts on the GPU against the oracle."""
import json
import os

import numpy as np
import pytest
import torch

transformers = pytest.importorskip("transformers")

VOCAB = ["<pad>", "<s>", "</s>", "<unk>", "|"] + list("abcdefghijklmnopqrstuvwxyz'")
CFG = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, feat_extract_norm="group",
           do_stable_layer_norm=False, vocab_size=len(VOCAB), conv_dim=(32,) * 7, conv_kernel=(10, 3, 3, 3, 3, 2, 2),
           conv_stride=(5, 2, 2, 2, 2, 2, 2), num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, pad_token_id=0)


@pytest.fixture(scope="module")
def checkpoint_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("w2v_ckpt"))
    torch.manual_seed(3)
    model = transformers.Wav2Vec2ForCTC(transformers.Wav2Vec2Config(**CFG)).eval()
    model.save_pretrained(d)
    with open(os.path.join(d, "vocab.json"), "w") as f:
        json.dump({t: i for i, t in enumerate(VOCAB)}, f)
    transformers.Wav2Vec2CTCTokenizer(os.path.join(d, "vocab.json")).save_pretrained(d)
    transformers.Wav2Vec2FeatureExtractor(return_attention_mask=False).save_pretrained(d)
    return d


def test_loader_builds_the_reference_module_layout(checkpoint_dir):
    from thunder_speech_amd.huggingface.compatibility import load_huggingface_checkpoint
    from thunder_speech_amd.huggingface.encoder import HuggingFaceEncoderAdapt
    from thunder_speech_amd.huggingface.transform import Wav2Vec2Preprocess
    m = load_huggingface_checkpoint(checkpoint_dir)
    assert not m.training and isinstance(m.encoder, HuggingFaceEncoderAdapt) and isinstance(m.audio_transform, Wav2Vec2Preprocess)
    assert m.encoder.mask_input is False and m.audio_transform.mask_input is False
    assert m.encoder_final_dimension == 64
    keys = m.state_dict().keys()
    assert "encoder.original_encoder.feature_extractor.conv_layers.0.conv.weight" in keys
    assert "decoder.2.weight" in keys and m.state_dict()["decoder.2.weight"].shape == (len(VOCAB), 64)
    # "|" is shown as a space; the CTC blank is the tokenizer's pad token
    assert m.text_transform.vocab.itos[4] == " " and m.text_transform.vocab.blank_token == "<pad>"
    assert m.text_transform.num_tokens == len(VOCAB)
    # the conv feature encoder is frozen, as in the reference
    assert not any(p.requires_grad for p in m.encoder.original_encoder.feature_extractor.parameters())


def test_unsupported_configurations_fail_loudly():
    from thunder_speech_amd.huggingface.encoder import HuggingFaceEncoderAdapt
    cfg = transformers.Wav2Vec2Config(**{**CFG, "hidden_act": "relu"})
    with pytest.raises(NotImplementedError, match="gelu"):
        HuggingFaceEncoderAdapt(transformers.Wav2Vec2Model(cfg))
    # the adapter behind the encoder is inference-only: building the module is fine, fine-tuning it says so
    HuggingFaceEncoderAdapt(transformers.Wav2Vec2Model(transformers.Wav2Vec2Config(**{**CFG, "add_adapter": True, "num_adapter_layers": 1})))
    enc = HuggingFaceEncoderAdapt(transformers.Wav2Vec2Model(transformers.Wav2Vec2Config(**CFG)))
    enc.train()
    with pytest.raises(RuntimeError):        # training mode runs on the HIP kernels too (huggingface/train.py); CPU tensors have no path at all
        enc(torch.zeros(1, 4000), torch.tensor([4000]))


def test_the_other_ctc_families_are_accepted_or_refused_by_name():
    """AutoModelForCTC families (huggingface/compatibility.py:77): hubert and data2vec-audio (tests/huggingface/test_module_huggingface.py:107-110)
    have a HIP path; a family whose layers this library has no kernels for says so when the adapter is built, not at the first forward."""
    from thunder_speech_amd.huggingface.encoder import HuggingFaceEncoderAdapt
    small = {k: v for k, v in CFG.items() if k not in ("feat_extract_norm", "do_stable_layer_norm")}
    HuggingFaceEncoderAdapt(transformers.HubertModel(transformers.HubertConfig(**{**CFG, "feat_proj_layer_norm": False})))
    HuggingFaceEncoderAdapt(transformers.Data2VecAudioModel(transformers.Data2VecAudioConfig(**{**small, "num_conv_pos_embeddings": 2})))
    with pytest.raises(NotImplementedError, match="wavlm"):
        HuggingFaceEncoderAdapt(transformers.WavLMModel(transformers.WavLMConfig(**CFG)))
    with pytest.raises(NotImplementedError, match="conv_pos_batch_norm"):
        HuggingFaceEncoderAdapt(transformers.HubertModel(transformers.HubertConfig(**{**CFG, "conv_pos_batch_norm": True})))


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["hubert", "hubert-large-style", "data2vec-audio", "unispeech", "unispeech-sat", "wav2vec2-adapter"])
def test_other_ctc_families_match_transformers_through_the_loader_path(family):
    """module_from_huggingface on a randomly initialised HubertForCTC / Data2VecAudioForCTC: encoder output and logits of the HIP path (fp32 mode)
    against the transformers forward pass of the same module."""
    from thunder_speech_amd.huggingface.compatibility import module_from_huggingface
    torch.manual_seed(3)
    small = {k: v for k, v in CFG.items() if k not in ("feat_extract_norm", "do_stable_layer_norm")}
    if family == "hubert":
        model = transformers.HubertForCTC(transformers.HubertConfig(**{**CFG, "feat_proj_layer_norm": False}))
    elif family == "hubert-large-style":
        model = transformers.HubertForCTC(transformers.HubertConfig(**{**CFG, "feat_extract_norm": "layer", "do_stable_layer_norm": True,
                                                                       "conv_bias": True, "feat_proj_layer_norm": True}))
    elif family == "wav2vec2-adapter":            # Wav2Vec2Adapter behind the encoder: projection to 48 channels, two conv + GLU layers; the head sits on 48
        model = transformers.Wav2Vec2ForCTC(transformers.Wav2Vec2Config(**{**CFG, "add_adapter": True, "num_adapter_layers": 2, "output_hidden_size": 48}))
    elif family == "unispeech":
        model = transformers.UniSpeechForCTC(transformers.UniSpeechConfig(**CFG))
    elif family == "unispeech-sat":
        model = transformers.UniSpeechSatForCTC(transformers.UniSpeechSatConfig(**{**CFG, "feat_extract_norm": "layer", "do_stable_layer_norm": True, "conv_bias": True}))
    else:
        model = transformers.Data2VecAudioForCTC(transformers.Data2VecAudioConfig(**{**small, "num_conv_pos_embeddings": 3, "conv_pos_kernel_size": 19}))
    model = model.eval()
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith(".bias"):
                v.copy_(0.1 * torch.randn_like(v))
    x = torch.randn(2, 6000)
    with torch.no_grad():
        want_h = model.base_model(x).last_hidden_state
        want_logits = model(x).logits
    fe = transformers.Wav2Vec2FeatureExtractor(return_attention_mask=False)
    m = module_from_huggingface(model, fe, None)
    m.encoder.precision = "fp32"
    m = m.cuda()
    with torch.no_grad():
        h, out_len = m.encoder(x.cuda(), torch.tensor([6000, 6000]).cuda())
    np.testing.assert_allclose(h.transpose(1, 2).cpu().numpy(), want_h.numpy(), atol=5e-4, rtol=1e-4)
    assert out_len.tolist() == [want_h.shape[1]] * 2 and m.encoder_final_dimension == want_h.shape[2]
    with torch.no_grad():
        logits = torch.nn.functional.linear(h.transpose(1, 2).cpu(), model.lm_head.weight, model.lm_head.bias)
    np.testing.assert_allclose(logits.numpy(), want_logits.numpy(), atol=2e-3, rtol=1e-3)


@pytest.mark.gpu
def test_logits_and_transcripts_match_the_oracle(checkpoint_dir):
    from oracle import w2v as ow
    from thunder_speech_amd.huggingface.compatibility import load_huggingface_checkpoint
    m = load_huggingface_checkpoint(checkpoint_dir)
    sd = {k: v.clone() for k, v in m.encoder.original_encoder.state_dict().items()}
    dec_w, dec_b = m.decoder[2].weight.detach().clone(), m.decoder[2].bias.detach().clone()
    m = m.cuda()
    g = torch.Generator().manual_seed(11)
    x = 0.1 * torch.randn(3, 12000, generator=g)
    lengths = torch.tensor([12000, 12000, 12000])
    with torch.no_grad():
        logits, out_len = m(x.cuda(), lengths.cuda())
        texts = m.predict(x.cuda())
    cfg = ow.W2VConfig(**{k: CFG[k] for k in ("conv_dim", "conv_kernel", "conv_stride", "hidden_size", "num_hidden_layers",
                                               "num_attention_heads", "intermediate_size", "num_conv_pos_embeddings",
                                               "num_conv_pos_embedding_groups")})
    xn = (x - x.mean(dim=1, keepdim=True)) / torch.sqrt(x.var(dim=1, keepdim=True) + 1e-7)      # Wav2Vec2Preprocess, mask_input=False
    h, _ = ow.forward(cfg, sd, xn)
    ref = (h @ dec_w.T + dec_b).transpose(1, 2)
    assert logits.shape == ref.shape and out_len.tolist() == [37, 37, 37]
    # the decoder kernel keeps its input in bf16 (as for QuartzNet): ~3 significant digits on the logits
    assert float((logits.float().cpu() - ref).abs().max()) <= 0.02 * max(1.0, float(ref.abs().max()))
    assert len(texts) == 3 and all(isinstance(t, str) for t in texts)
