"""Checkpoint import, offline: the NeMo configs the reference's own tests hold
(tests/nemo_config_samples/*.yaml, copied as data into tests/golden/nemo_config_samples/) and synthetic `.nemo` archives
written with NeMo's key naming.  No network, no real weights."""
import io
import tarfile
from pathlib import Path

import pytest
import torch
import yaml

from thunder_speech_amd.quartznet.blocks import QuartznetEncoder
from thunder_speech_amd.quartznet.compatibility import (QuartznetCheckpoint, fix_encoder_name,
                                                        load_components_from_quartznet_config,
                                                        load_quartznet_checkpoint)

SAMPLES = Path(__file__).parent / "golden" / "nemo_config_samples"


@pytest.mark.parametrize("name", ["QuartzNet5x5LS-En.yaml", "QuartzNet15x5Base-En.yaml", "QuartzNet15x5NR-En.yaml"])
def test_components_from_reference_sample_configs(name):
    """Mirror of the reference's test_create_from_manifest (tests/quartznet/test_compatibility_qn.py:30-54), minus the
    forward pass: module tree / state-dict compatibility with the plain constructors."""
    encoder, fb, text_tfm = load_components_from_quartznet_config(SAMPLES / name)
    ref = QuartznetEncoder() if "Net5x5" in name else QuartznetEncoder(repeat_blocks=3)
    ref.load_state_dict(encoder.state_dict(), strict=True)
    assert len(encoder.state_dict()) == (225 if "Net5x5" in name else 635)
    assert fb[2].layer[0].fb.shape[1] == 64                     # 64 mel bins
    assert text_tfm.num_tokens == 29 and text_tfm.vocab.blank_idx == 28   # 28 labels + blank


def test_fix_encoder_name_rules():
    assert fix_encoder_name("encoder.encoder.3.mconv.1.conv.weight") == "3.mconv.1.conv.weight"
    assert fix_encoder_name("encoder.encoder.3.mconv.2.running_mean") == "3.mconv.2.layer.0.running_mean"
    assert fix_encoder_name("encoder.encoder.3.res.0.0.conv.weight") == "3.res.0.conv.weight"
    assert fix_encoder_name("encoder.encoder.3.res.0.1.num_batches_tracked") == "3.res.1.layer.0.num_batches_tracked"


def _nemo_name(key: str) -> str:
    """Inverse of the renaming rule, for writing a synthetic archive: our key -> NeMo's."""
    parts = key.split(".")
    if "res" in parts:
        r = parts.index("res")
        parts = parts[:r + 1] + ["0"] + parts[r + 1:]
    parts = [p for i, p in enumerate(parts) if not (p == "layer" and parts[i + 1] == "0") and not (p == "0" and i > 0 and parts[i - 1] == "layer")]
    return "encoder.encoder." + ".".join(parts)


def test_synthetic_nemo_archive_round_trip(tmp_path):
    cfg = yaml.safe_load(open(SAMPLES / "QuartzNet5x5LS-En.yaml"))
    src = QuartznetEncoder()
    g = torch.Generator().manual_seed(0)
    sd = {k: (torch.randn(v.shape, generator=g) if v.is_floating_point() else v.clone()) for k, v in src.state_dict().items()}
    weights = {_nemo_name(k): v for k, v in sd.items()}
    assert all(fix_encoder_name(k) in sd for k in weights)                         # the inverse really is the inverse
    dec_w, dec_b = torch.randn(29, 1024, 1, generator=g), torch.randn(29, generator=g)
    weights["decoder.decoder_layers.0.weight"], weights["decoder.decoder_layers.0.bias"] = dec_w, dec_b
    ckpt = tmp_path / "model_weights.ckpt"
    torch.save(weights, ckpt)
    cfg_path = tmp_path / "model_config.yaml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    nemo = tmp_path / "synthetic.nemo"
    with tarfile.open(nemo, "w:gz") as tar:
        tar.add(ckpt, arcname="./model_weights.ckpt")
        tar.add(cfg_path, arcname="./model_config.yaml")
    module = load_quartznet_checkpoint(str(nemo))
    assert not module.training and module.encoder_final_dimension == 1024
    got = module.encoder.state_dict()
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    assert torch.equal(module.decoder.weight, dec_w) and torch.equal(module.decoder.bias, dec_b)
    assert module.text_transform.num_tokens == 29


def test_missing_checkpoint_file_is_reported():
    with pytest.raises(FileNotFoundError):
        load_quartznet_checkpoint(QuartznetCheckpoint.QuartzNet5x5LS_En, save_folder="/nonexistent")


def test_citrinet_config_and_vocab():
    from thunder_speech_amd.citrinet.compatibility import fix_vocab
    assert fix_vocab(["##ing", "the", "##s"]) == ["ing", "▁the", "s"]           # citrinet/compatibility.py:114-130


def test_synthetic_citrinet_nemo_archive_round_trip(tmp_path):
    """load_citrinet_checkpoint end to end (reference citrinet/compatibility.py:133-176) on a synthetic archive with NeMo's
    layout: model_config.yaml (encoder.jasper list with stem / body / head entries, preprocessor, decoder.vocabulary in NeMo's
    "##" word-piece spelling), model_weights.ckpt with NeMo's key naming, and a sentencepiece tokenizer.model -- the one the
    reference's own tests hold (tests/nemo_config_samples/example_tokenizer.model)."""
    import sentencepiece as spm
    from thunder_speech_amd.citrinet.blocks import CitrinetEncoder
    from thunder_speech_amd.citrinet.compatibility import load_citrinet_checkpoint
    tok_path = SAMPLES / "example_tokenizer.model"
    sp = spm.SentencePieceProcessor(model_file=str(tok_path))
    pieces = [sp.id_to_piece(i) for i in range(sp.get_piece_size())]
    nemo_vocab = [("##" + p) if not p.startswith("▁") else p[1:] for p in pieces]       # inverse of fix_vocab
    body = [dict(filters=32, kernel=[5], stride=[2]), dict(filters=48, kernel=[7], stride=[1])]
    cfg = {"encoder": {"jasper": [dict(filters=256, kernel=[5], stride=[1])] + body + [dict(filters=640, kernel=[41], stride=[1])]},
           "preprocessor": dict(sample_rate=16000, window_size=0.025, window_stride=0.01, n_fft=512, features=80, dither=1e-5),
           "decoder": {"vocabulary": nemo_vocab}}
    src = CitrinetEncoder(filters=[32, 48], kernel_sizes=[5, 7], strides=[2, 1])
    g = torch.Generator().manual_seed(1)
    sd = {k: (torch.randn(v.shape, generator=g) if v.is_floating_point() else v.clone()) for k, v in src.state_dict().items()}
    weights = {_nemo_name(k): v for k, v in sd.items()}
    assert all(fix_encoder_name(k) in sd for k in weights)
    n_tok = len(pieces) + 1                                                             # + blank
    dec_w, dec_b = torch.randn(n_tok, 640, 1, generator=g), torch.randn(n_tok, generator=g)
    weights["decoder.decoder_layers.0.weight"], weights["decoder.decoder_layers.0.bias"] = dec_w, dec_b
    torch.save(weights, tmp_path / "model_weights.ckpt")
    (tmp_path / "model_config.yaml").write_text(yaml.safe_dump(cfg))
    nemo = tmp_path / "synthetic_citrinet.nemo"
    with tarfile.open(nemo, "w:gz") as tar:
        tar.add(tmp_path / "model_weights.ckpt", arcname="./model_weights.ckpt")
        tar.add(tmp_path / "model_config.yaml", arcname="./model_config.yaml")
        tar.add(tok_path, arcname="./abc123_tokenizer.model")                           # NeMo prefixes the file with a hash
    module = load_citrinet_checkpoint(str(nemo), augment_params=dict(num_time_masks=2, mask_time_width=30, dropout=0.1))
    assert not module.training and module.encoder_final_dimension == 640
    got = module.encoder.state_dict()
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    assert torch.equal(module.decoder.weight, dec_w) and module.text_transform.num_tokens == n_tok
    assert len(module.audio_transform) == 5                                             # augment_params reached the front end ...
    assert any(isinstance(m, torch.nn.Dropout) and m.p == 0.1 for m in module.encoder.modules())   # ... and the encoder
    assert module.audio_transform[2].layer[0].fb.shape[1] == 80 and module.audio_transform[1].win_length == 400
    # the text side tokenises with the archive's sentencepiece model and round-trips through the vocabulary
    ids, lens = module.text_transform.encode(["hello world"])
    assert module.text_transform.decode_prediction(ids, remove_repeated=False) == [" hello world"]      # "▁" -> " " (transform.py:117)
    assert module.text_transform.vocab.itos[:5] == pieces[:5]                           # fix_vocab restored the sentencepiece spelling
