"""Oracle front end vs fixtures produced by the real reference FilterbankFeatures (CPU only)."""
import numpy as np
import pytest
import torch

from oracle import frontend as fe
from oracle import primitives as prim


@pytest.mark.parametrize("tag,kw", [("qn", {}), ("cn", dict(n_window_size=400, nfilt=80))])
def test_frontend_stages_match_reference(golden, tag, kw):
    g = golden(f"frontend_{tag}.npz")
    cfg = fe.FrontendConfig(**kw)
    st = fe.filterbank_features(torch.from_numpy(g["x"]), torch.from_numpy(g["lengths"]), cfg, return_stages=True)
    assert np.array_equal(st["lengths"].numpy(), g["feat_lengths"])
    assert st["features"].shape == g["features"].shape == (3, cfg.nfilt, 4000 // 160 + 1)
    np.testing.assert_allclose(st["preemph"].numpy(), g["preemph"], atol=1e-7)
    np.testing.assert_allclose(st["power"].numpy(), g["power"], atol=2e-4 * g["power"].max())
    np.testing.assert_allclose(st["logmel"].numpy(), g["logmel"], atol=2e-4)
    # the reference's own CPU/GPU tolerance for the filterbank is 1e-3 (tests/quartznet/test_transform_qn.py:307)
    np.testing.assert_allclose(st["features"].numpy(), g["features"], atol=1e-3)
    # frames beyond the length are exactly zero
    for b, n in enumerate(g["feat_lengths"]):
        assert np.all(st["features"].numpy()[b, :, n:] == 0)


def test_fp64_front_end_agrees_with_fp32_reference(golden):
    g = golden("frontend_qn.npz")
    f64, _ = fe.filterbank_features(torch.from_numpy(g["x"]), torch.from_numpy(g["lengths"]), dtype=torch.float64)
    np.testing.assert_allclose(f64.numpy(), g["features"], atol=1e-3)


@pytest.mark.parametrize("n_mels,n_fft", [(64, 512), (80, 512), (40, 400)])
def test_mel_filterbank_vs_independent_implementation(n_mels, n_fft):
    """Filterbank VALUES are unpinned by the reference's tests; cross-check the slaney restatement
    with transformers' independent implementation."""
    ta = pytest.importorskip("transformers.audio_utils")
    ours = fe.slaney_mel_filterbank(n_fft // 2 + 1, n_mels, 16000)
    theirs = ta.mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=0.0,
                                max_frequency=8000.0, sampling_rate=16000, norm="slaney", mel_scale="slaney")
    np.testing.assert_allclose(ours, np.asarray(theirs).T, atol=1e-6)
    assert ours.shape == (n_mels, n_fft // 2 + 1) and np.isfinite(ours).all()


def test_mel_filterbank_matches_fixture(golden):
    g = golden("frontend_qn.npz")
    np.testing.assert_allclose(fe.slaney_mel_filterbank(257, 64, 16000), g["mel_fb"], atol=1e-7)


def test_power_spectrum_shape_and_lengths():
    # reference: tests/quartznet/test_transform_qn.py:179-189  [B, 1+n_fft//2, 1+T//hop]
    cfg = fe.FrontendConfig()
    x = torch.randn(2, 1234)
    p = fe.power_spectrum(x, cfg)
    assert p.shape == (2, 257, 1234 // 160 + 1)
    assert fe.feature_lengths(torch.tensor([1234.0, 159.0, 160.0]), 160).tolist() == [8, 1, 2]


def test_melscale_finite_on_zeros():
    # reference: tests/quartznet/test_transform_qn.py:239-245
    cfg = fe.FrontendConfig()
    out = fe.log_mel(torch.zeros(1, 257, 5), cfg)
    assert torch.isfinite(out).all() and torch.allclose(out, torch.full_like(out, float(np.log(2.0 ** -24))))


def test_normalize_primitives_match_reference(golden):
    g = golden("primitives.npz")
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lengths"])
    mask = prim.lengths_to_mask(lens, x.shape[-1]).unsqueeze(1)
    np.testing.assert_allclose(prim.masked_normalize(x, mask, 1e-5).numpy(), g["masked"], atol=1e-5)
    np.testing.assert_allclose(prim.unmasked_normalize(x[:, 0], 1e-7).numpy(), g["unmasked"], atol=1e-5)
    np.testing.assert_allclose(prim.wav2vec2_preprocess(x[:, 0], lens, True)[0].numpy(), g["w2v_masked"], atol=1e-5)
    np.testing.assert_allclose(prim.wav2vec2_preprocess(x[:, 0], lens, False)[0].numpy(), g["w2v_unmasked"], atol=1e-5)


def test_quirk_a1_padded_frames_inflate_std():
    """sigma^2 = (sum_valid (x-mu)^2 + (T-N) mu^2) / N, not the textbook masked variance."""
    x = torch.tensor([[[1.0, 3.0, 100.0, 100.0]]])
    mask = torch.tensor([[[True, True, False, False]]])
    out = prim.masked_normalize(x, mask, 0.0)
    mu = 2.0
    sigma = np.sqrt(((1 - mu) ** 2 + (3 - mu) ** 2 + 2 * mu ** 2) / 2)
    np.testing.assert_allclose(out[0, 0, :2].numpy(), [(1 - mu) / sigma, (3 - mu) / sigma], rtol=1e-6)
    assert out[0, 0, 2:].abs().sum() == 0


def test_lengths_to_mask_truth_table():
    # the reference's literal known-answer vector: tests/test_blocks.py:56-68
    m = prim.lengths_to_mask(torch.tensor([1, 3, 5]), 5)
    expected = torch.tensor([[1, 0, 0, 0, 0], [1, 1, 1, 0, 0], [1, 1, 1, 1, 1]], dtype=torch.bool)
    assert torch.equal(m, expected)
    assert torch.equal(prim.lengths_to_mask(torch.tensor([1.9, 3.0]), 4),
                       torch.tensor([[1, 0, 0, 0], [1, 1, 1, 0]], dtype=torch.bool))
