"""GPU ports of the reference's own parameter sweeps for the modules of the hot path, checked against the oracle instead of only for
shapes: QuartznetBlock over (channels, repeat, kernel, stride 1-3, dilation 1-2, residual, separable) -- reference
tests/quartznet/test_blocks_qn.py:158-243 -- and the five front-end stage modules called on their own -- reference
tests/quartznet/test_transform_qn.py:130-260.  (The reference's script / device-move variants have no counterpart: custom HIP ops are
not TorchScript-able and there is no CPU path)"""
import math

import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis.strategies import booleans, floats, integers, lists, none, one_of

from oracle import frontend as ofe
from oracle import tcs as otcs
from oracle.primitives import bf16_round, lengths_to_mask, masked_normalize

pytestmark = pytest.mark.gpu
SWEEP = settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)


def _block_case(in_channels, out_channels, repeat, kernel_size, stride, dilation, residual, separable, t=337, batch=4, seed=0):
    from thunder_speech_amd.quartznet.blocks import QuartznetBlock
    try:
        block = QuartznetBlock(in_channels, out_channels, repeat=repeat, kernel_size=kernel_size, stride=stride, dilation=dilation,
                               residual=residual, separable=separable)
    except ValueError:
        return None                                    # stride > 1 with dilation > 1: the reference refuses it too
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in block.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(1 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
    sd = {k: v.detach().clone() for k, v in block.state_dict().items()}
    spec = otcs.BlockSpec(in_channels, out_channels, repeat=repeat, kernel=kernel_size[0], stride=stride[0], dilation=dilation[0],
                          residual=residual, separable=separable)
    x = bf16_round(torch.randn(batch, in_channels, t, generator=g))
    lens = torch.randint(10, t, (batch,), generator=g)
    lens[0] = t
    ref, ref_len = otcs.block_forward(spec, sd, "", x, lens, emulate_bf16=True)
    out, out_len = block.cuda().eval()(x.cuda(), lens.cuda())
    assert out.shape[0] == batch and out.shape[1] == out_channels                 # what the reference's sweep asserts
    assert torch.equal(out_len.cpu(), ref_len)
    got = out.float().cpu()
    assert got.shape == ref.shape
    scale = max(1.0, float(ref.abs().max()))
    for b, n in enumerate(ref_len.tolist()):                                        # valid frames: bf16 parity with the oracle
        assert float((got[b, :, : int(n)] - ref[b, :, : int(n)]).abs().max()) <= 0.02 * scale
    return True


@SWEEP
@given(in_channels=integers(16, 32), out_channels=integers(16, 32), repeat=integers(1, 4),
       kernel_size=lists(integers(11, 33).filter(lambda x: x % 2 == 1), min_size=1, max_size=1),
       stride=lists(integers(1, 3), min_size=1, max_size=1), dilation=lists(integers(1, 2), min_size=1, max_size=1),
       residual=booleans(), separable=booleans())
def test_quartznet_block_combinations_match_oracle(**kwargs):
    _block_case(**kwargs)


@pytest.mark.parametrize("kw", [
    dict(in_channels=16, out_channels=24, repeat=2, kernel_size=[11], stride=[1], dilation=[1], residual=True, separable=False),   # dense K > 1
    dict(in_channels=20, out_channels=20, repeat=3, kernel_size=[13], stride=[2], dilation=[1], residual=True, separable=False),   # dense, strided
    dict(in_channels=16, out_channels=32, repeat=1, kernel_size=[15], stride=[1], dilation=[2], residual=False, separable=False),  # dense, dilated
    dict(in_channels=24, out_channels=24, repeat=2, kernel_size=[11], stride=[3], dilation=[1], residual=True, separable=True),    # depthwise stride 3
    dict(in_channels=17, out_channels=31, repeat=4, kernel_size=[33], stride=[3], dilation=[1], residual=False, separable=False),  # dense stride 3
    dict(in_channels=32, out_channels=16, repeat=2, kernel_size=[21], stride=[2], dilation=[1], residual=True, separable=True),    # stride 2 (fused kernel)
])
def test_quartznet_block_forms_without_a_fused_kernel_match_oracle(kw):
    assert _block_case(**kw) is True


# ------------------------------------------------------------------------------------------------ front-end stages on their own
@SWEEP
@given(preemph=floats(min_value=0.000001, max_value=0.999999999))
def test_preemph_filter_matches_oracle(preemph):
    from thunder_speech_amd.quartznet.transform import PreEmphasisFilter
    x = torch.randn(10, 1337, generator=torch.Generator().manual_seed(1))
    out = PreEmphasisFilter(preemph)(x.cuda()).cpu()
    assert out.shape == x.shape
    assert torch.allclose(out, ofe.preemphasis(x, preemph), atol=1e-6)
    assert not torch.allclose(out, x)


def test_preemph_filter_zero_gain_is_the_identity():
    from thunder_speech_amd.quartznet.transform import PreEmphasisFilter
    x = torch.randn(10, 1337)
    assert torch.equal(PreEmphasisFilter(0.0)(x.cuda()).cpu(), x)


@SWEEP
@given(n_window_size=integers(min_value=16, max_value=128), n_window_stride=integers(min_value=8, max_value=64),
       n_fft=one_of(none(), integers(min_value=128, max_value=256)))
def test_powerspectrum_matches_oracle(n_window_size, n_window_stride, n_fft):
    from thunder_speech_amd.quartznet.transform import PowerSpectrum
    spec = PowerSpectrum(n_window_size=n_window_size, n_window_stride=n_window_stride, n_fft=n_fft)
    x = torch.randn(10, 1337, generator=torch.Generator().manual_seed(2))
    lens = torch.Tensor([1337] * 9 + [700])
    out, out_lens = spec.cuda()(x.cuda(), lens.cuda())
    assert out.shape == (10, 1 + spec.n_fft // 2, 1 + 1337 // spec.hop_length)
    assert torch.equal(out_lens.cpu(), (torch.floor(lens / spec.hop_length) + 1).long())
    cfg = ofe.FrontendConfig(n_window_size=n_window_size, n_window_stride=n_window_stride, n_fft=spec.n_fft)
    ref = ofe.power_spectrum(x, cfg)
    assert float((out.cpu() - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max()))


@SWEEP
@given(n_window_size=integers(max_value=0), n_window_stride=integers(max_value=0))
def test_powerspec_raises(n_window_size, n_window_stride):
    from thunder_speech_amd.quartznet.transform import PowerSpectrum
    with pytest.raises(ValueError):
        PowerSpectrum(n_window_size=n_window_size, n_window_stride=n_window_stride)


@SWEEP
@given(sample_rate=integers(min_value=8000, max_value=9000), n_fft=integers(min_value=500, max_value=512), nfilt=integers(min_value=60, max_value=64))
def test_melscale_matches_oracle_and_guards_zero(sample_rate, n_fft, nfilt):
    from thunder_speech_amd.quartznet.transform import MelScale
    mel = MelScale(sample_rate=sample_rate, n_fft=n_fft, nfilt=nfilt).cuda()
    n_freq = 1 + n_fft // 2
    x = torch.randn(10, n_freq, 137, generator=torch.Generator().manual_seed(3)).abs()
    out = mel(x.cuda()).cpu()
    assert out.shape == (10, nfilt, 137) and torch.isfinite(out).all()
    fb = torch.from_numpy(ofe.slaney_mel_filterbank(n_freq, nfilt, sample_rate))
    ref = torch.log(torch.matmul(fb.unsqueeze(0), x) + ofe.LOG_FLOOR)
    assert float((out - ref).abs().max()) <= 2e-4
    zero = mel(torch.zeros(2, n_freq, 5, device="cuda")).cpu()
    assert torch.isfinite(zero).all() and torch.allclose(zero, torch.full_like(zero, math.log(ofe.LOG_FLOOR)))


def test_filterbank_features_without_log_scale_matches_oracle():
    """MelScale(log_scale=False) (reference transform.py:213, :253): the plain filterbank product goes into the normaliser.  The fused front
    end has the log inside its kernel, so this configuration runs the module chain stage by stage -- it must not raise, and it must
    match the oracle's stages with the log left out."""
    from thunder_speech_amd.quartznet.transform import FilterbankFeatures
    fbk = FilterbankFeatures().cuda().eval()
    fbk[2].layer[0].log_scale = False
    g = torch.Generator().manual_seed(9)
    wav = 0.1 * torch.randn(3, 8000, generator=g)
    lengths = torch.tensor([8000, 6000, 1234])
    for b, n in enumerate(lengths.tolist()):
        wav[b, n:] = 0
    feats, flen = fbk(wav.cuda(), lengths.cuda())
    cfg = ofe.FrontendConfig()
    power = ofe.power_spectrum(ofe.preemphasis(wav, cfg.preemph), cfg)
    fb = torch.from_numpy(ofe.slaney_mel_filterbank(cfg.n_freqs, cfg.nfilt, cfg.sample_rate))
    mel = torch.matmul(fb.unsqueeze(0), power)
    want_len = ofe.feature_lengths(lengths, cfg.n_window_stride)
    ref = masked_normalize(mel, lengths_to_mask(want_len, mel.shape[-1]).unsqueeze(1), div_guard=1e-5)
    assert torch.equal(flen.cpu(), want_len)
    got = feats.float().cpu()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) <= 2e-3 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("lens", [[137, 100, 1, 64], [137.0, 99.5, 20.0, 137.0]])
def test_feature_batch_normalizer_matches_oracle(lens):
    from thunder_speech_amd.quartznet.transform import FeatureBatchNormalizer
    x = torch.randn(4, 64, 137, generator=torch.Generator().manual_seed(4)) * 3 + 1
    lengths = torch.tensor(lens)
    out, out_len = FeatureBatchNormalizer()(x.cuda(), lengths.cuda())
    assert torch.equal(out_len.cpu(), lengths)
    mask = lengths_to_mask(lengths, 137).unsqueeze(1)
    ref = masked_normalize(x, mask, div_guard=1e-5)
    assert float((out.cpu() - ref).abs().max()) <= 1e-4


def test_dither_audio_is_the_oracles_noise_in_training_and_the_identity_in_eval(monkeypatch):
    from oracle import philox as ph
    from thunder_speech_amd import rng
    from thunder_speech_amd.quartznet.transform import DitherAudio
    x = torch.randn(3, 1001, generator=torch.Generator().manual_seed(5))
    d = DitherAudio(dither=1e-2)
    assert torch.equal(d.eval()(x.cuda()).cpu(), x)
    monkeypatch.setattr(rng, "next_seed", lambda: 4242)
    out = d.train()(x.cuda()).cpu()
    noise = torch.from_numpy(np.stack([ph.dither_noise(4242, b, x.shape[1]) for b in range(x.shape[0])]))
    assert torch.allclose(out, x + 1e-2 * noise, atol=1e-6)
