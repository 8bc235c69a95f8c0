"""`.nemo` archive -> load_quartznet_checkpoint / load_citrinet_checkpoint -> .cuda() -> HIP logits vs the oracle fed the SAME state
dict (reference quartznet/compatibility.py:127-201, citrinet/compatibility.py:114-176).  The archives are synthetic
(real weights need the network) but carry NeMo's file layout and key naming; the weights are the oracle's calibrated synthetic ones,
so the forward pass is numerically meaningful."""
import tarfile
from pathlib import Path

import numpy as np
import pytest
import torch
import yaml

from oracle import decode as odec
from oracle import frontend as ofe
from oracle import tcs as otcs
from oracle.primitives import bf16_round
from tests.test_checkpoint_compat import SAMPLES, _nemo_name

pytestmark = pytest.mark.gpu


def _rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, dtype=np.float64)))))


def _write_nemo(tmp_path, name, cfg, weights, extra=()):
    torch.save(weights, tmp_path / "model_weights.ckpt")
    (tmp_path / "model_config.yaml").write_text(yaml.safe_dump(cfg))
    nemo = tmp_path / name
    with tarfile.open(nemo, "w:gz") as tar:
        tar.add(tmp_path / "model_weights.ckpt", arcname="./model_weights.ckpt")
        tar.add(tmp_path / "model_config.yaml", arcname="./model_config.yaml")
        for path, arc in extra:
            tar.add(path, arcname=arc)
    return nemo


def _compare(module, ref, emu, wav, vocab_strings):
    """HIP logits vs fp32 oracle logits `ref`: no less accurate than the oracle's own bf16-ordered evaluation `emu` of the same
    network (the tolerance statement of DESIGN.md section 4: activations are stored in bf16); greedy frames equal wherever the oracle's
    margin exceeds 6 sigma of the error; the strings predict() returns are the reference decode of the module's own argmax."""
    lengths = torch.full((wav.shape[0],), wav.shape[1])
    with torch.no_grad():
        logits, _ = module(wav.cuda(), lengths.cuda())
        strings = module.predict(wav.cuda())
    got = logits.float().cpu().numpy()
    r = ref.numpy()
    scale = float(np.abs(r).max())
    assert np.isfinite(got).all()
    assert float(np.abs(got - r).max()) <= 0.08 * scale, f"max err {np.abs(got - r).max()} scale {scale}"
    assert _rms(got - r) <= 1.25 * _rms(emu.numpy() - r) + 1e-3 * scale
    noise = _rms(got - r)
    top2 = np.sort(r, axis=1)[:, -2:, :]
    decided = (top2[:, 1] - top2[:, 0]) > 6 * noise
    assert decided.mean() > 0.25          # random weights: many frames are near-ties
    assert np.array_equal(got.argmax(1)[decided], r.argmax(1)[decided])
    assert strings == vocab_strings(got.argmax(1))


def test_quartznet_nemo_archive_to_hip_logits(tmp_path):
    from thunder_speech_amd.quartznet.compatibility import fix_encoder_name, load_quartznet_checkpoint
    cfg = yaml.safe_load(open(SAMPLES / "QuartzNet5x5LS-En.yaml"))
    arch = otcs.quartznet_arch(repeat_blocks=1)
    sd = otcs.synth_encoder_state(arch, seed=11, calibrate=True)
    dsd = otcs.synth_decoder_state(1024, 29, seed=12, gain=4.0)
    weights = {_nemo_name(k): v for k, v in sd.items()}
    assert all(fix_encoder_name(k) in sd for k in weights)
    weights["decoder.decoder_layers.0.weight"], weights["decoder.decoder_layers.0.bias"] = dsd["weight"], dsd["bias"]
    module = load_quartznet_checkpoint(str(_write_nemo(tmp_path, "qn5x5.nemo", cfg, weights))).cuda()
    assert not module.training
    rng = np.random.Generator(np.random.PCG64(21))
    wav = torch.from_numpy((0.1 * rng.standard_normal((3, 24000))).astype(np.float32))
    lengths = torch.full((3,), 24000)
    feats, fl = ofe.filterbank_features(wav, lengths)
    enc, _ = otcs.encoder_forward(arch, sd, feats, fl)
    ref = otcs.conv1d_decoder_forward(dsd, enc)
    enc_e, _ = otcs.encoder_forward(arch, sd, bf16_round(feats), fl, emulate_bf16=True)
    emu = otcs.conv1d_decoder_forward(dsd, enc_e, emulate_bf16=True)
    labels = cfg["labels"] if "labels" in cfg else cfg["decoder"]["vocabulary"]
    vocab = odec.Vocab(list(labels))
    _compare(module, ref, emu, wav, lambda ids: odec.decode_prediction(ids, vocab))


def test_citrinet_nemo_archive_to_hip_logits(tmp_path):
    import sentencepiece as spm
    from thunder_speech_amd.citrinet.compatibility import load_citrinet_checkpoint
    from thunder_speech_amd.quartznet.compatibility import fix_encoder_name
    tok_path = SAMPLES / "example_tokenizer.model"
    sp = spm.SentencePieceProcessor(model_file=str(tok_path))
    pieces = [sp.id_to_piece(i) for i in range(sp.get_piece_size())]
    nemo_vocab = [("##" + p) if not p.startswith("▁") else p[1:] for p in pieces]
    filters, kernels, strides = [64, 128], [5, 7], [2, 1]
    body = [dict(filters=f, kernel=[k], stride=[s]) for f, k, s in zip(filters, kernels, strides)]
    cfg = {"encoder": {"jasper": [dict(filters=256, kernel=[5], stride=[1])] + body + [dict(filters=640, kernel=[41], stride=[1])]},
           "preprocessor": dict(sample_rate=16000, window_size=0.025, window_stride=0.01, n_fft=512, features=80, dither=1e-5),
           "decoder": {"vocabulary": nemo_vocab}}
    arch = otcs.citrinet_arch(filters, kernels, strides, feat_in=80)
    sd = otcs.synth_encoder_state(arch, seed=13, calibrate=True)
    n_tok = len(pieces) + 1
    dsd = otcs.synth_decoder_state(640, n_tok, seed=14, gain=4.0)
    weights = {_nemo_name(k): v for k, v in sd.items()}
    assert all(fix_encoder_name(k) in sd for k in weights)
    weights["decoder.decoder_layers.0.weight"], weights["decoder.decoder_layers.0.bias"] = dsd["weight"], dsd["bias"]
    nemo = _write_nemo(tmp_path, "citrinet.nemo", cfg, weights, extra=[(tok_path, "./abc123_tokenizer.model")])
    module = load_citrinet_checkpoint(str(nemo)).cuda()
    assert not module.training and module.encoder_final_dimension == 640
    rng = np.random.Generator(np.random.PCG64(22))
    wav = torch.from_numpy((0.1 * rng.standard_normal((2, 32000))).astype(np.float32))
    lengths = torch.full((2,), 32000)
    feats, fl = ofe.filterbank_features(wav, lengths, ofe.FrontendConfig(n_window_size=400, nfilt=80))
    enc, _ = otcs.encoder_forward(arch, sd, feats, fl)
    ref = otcs.conv1d_decoder_forward(dsd, enc)
    enc_e, _ = otcs.encoder_forward(arch, sd, bf16_round(feats), fl, emulate_bf16=True)
    emu = otcs.conv1d_decoder_forward(dsd, enc_e, emulate_bf16=True)
    tt = module.text_transform
    _compare(module, ref, emu, wav, lambda ids: tt.decode_prediction(torch.from_numpy(np.asarray(ids))))
