"""GPU parity: Citrinet blocks (squeeze-excite launch sequence) through the module mirror / C ABI vs the CPU oracle and the
reference fixtures."""
import numpy as np
import pytest
import torch

from oracle import tcs as otcs
from oracle.primitives import bf16_round

pytestmark = pytest.mark.gpu


def _block(spec, sd):
    from thunder_speech_amd.citrinet.blocks import CitrinetBlock
    blk = CitrinetBlock(spec.in_ch, spec.out_ch, repeat=spec.repeat, kernel_size=(spec.kernel,), stride=(spec.stride,),
                        dilation=(spec.dilation,), residual=spec.residual, separable=spec.separable)
    blk.load_state_dict(sd, strict=True)
    return blk.cuda().eval()


def test_squeeze_excite_matches_reference_fixture(golden):
    from thunder_speech_amd.citrinet.blocks import SqueezeExcite
    g = golden("conv_se.npz")
    se = SqueezeExcite(16, 8)
    se.load_state_dict({"fc.0.weight": torch.from_numpy(g["se_w1"]), "fc.2.weight": torch.from_numpy(g["se_w2"])})
    x = torch.from_numpy(g["se_x"])
    y = se.cuda()(x.cuda()).cpu()
    ref = otcs.squeeze_excite(bf16_round(x), torch.from_numpy(g["se_w1"]), torch.from_numpy(g["se_w2"]))
    scale = float(np.abs(g["se_y"]).max())
    assert float((y - ref).abs().max()) <= 0.01 * scale                  # bf16 in, bf16 out
    assert float((y - torch.from_numpy(g["se_y"])).abs().max()) <= 0.02 * scale


@pytest.mark.parametrize("cin,cout,repeat,k,stride,t,lens,res", [
    (16, 32, 3, 7, 1, 48, [48, 31, 6], True),          # the reference-fixture geometries (tests/golden/blocks.npz cn_s1 / cn_s2)
    (32, 32, 2, 9, 2, 51, [51, 30, 10], True),
    (128, 128, 3, 13, 1, 300, [300, 211, 97], True),   # tail-zero fast kernels (c_in % 64 == 0)
    (128, 256, 2, 25, 2, 301, [301, 150, 33], True),   # stride on the last repeat, residual stride 2
    (64, 320, 1, 41, 1, 140, [140, 77], False),        # head geometry: no residual
])
def test_citrinet_block_matches_oracle(cin, cout, repeat, k, stride, t, lens, res):
    spec = otcs.BlockSpec(cin, cout, repeat=repeat, kernel=k, stride=stride, residual=res, family="citrinet")
    sd = {key[2:]: v for key, v in otcs.synth_encoder_state([spec], seed=3).items()}
    g = torch.Generator().manual_seed(11)
    x = bf16_round(torch.randn(len(lens), cin, t, generator=g))
    lengths = torch.tensor(lens)
    ref, ref_len = otcs.block_forward(spec, sd, "", x, lengths, emulate_bf16=True)
    ref32, _ = otcs.block_forward(spec, sd, "", x, lengths, emulate_bf16=False)
    y, yl = _block(spec, sd)(x.cuda(), lengths.cuda())
    torch.cuda.synchronize()
    assert torch.equal(yl.cpu(), ref_len)
    got = y.float().cpu()
    assert got.shape == ref.shape
    scale = max(1.0, float(ref.abs().max()))
    # every frame, padded ones included: predict() decodes all of them (quirk A2 / A9)
    assert float((got - ref).abs().max()) <= 0.02 * scale
    assert float((got - ref32).abs().max()) <= 0.05 * scale


def test_citrinet_float_lengths_and_internal_chain():
    """Two blocks chained through EncoderSequential: the first is internal (arena buffers, zeroed tails), lengths float (A5)."""
    from thunder_speech_amd.citrinet.blocks import CitrinetEncoder
    arch = otcs.citrinet_arch(filters=[64, 64], kernel_sizes=[11, 13], strides=[1, 2], feat_in=32)
    sd = otcs.synth_encoder_state(arch, seed=5)
    # the reference (and the mirror) hard-code a 256-channel stem and a 640-channel head: citrinet/blocks.py:209-216, 244-254
    enc = CitrinetEncoder(filters=[64, 64], kernel_sizes=[11, 13], strides=[1, 2], feat_in=32)
    ref_keys = set(enc.state_dict().keys())
    assert ref_keys == set(sd.keys())
    enc.load_state_dict(sd, strict=True)
    enc = enc.cuda().eval()
    g = torch.Generator().manual_seed(2)
    x = bf16_round(torch.randn(2, 32, 200, generator=g))
    lengths = torch.tensor([200.0, 131.0])
    ref, ref_len = otcs.encoder_forward(arch, sd, x, lengths, emulate_bf16=True)
    y, yl = enc(x.cuda(), lengths.cuda())
    assert yl.dtype == lengths.dtype and torch.equal(yl.cpu(), ref_len)
    scale = max(1.0, float(ref.abs().max()))
    assert float((y.float().cpu() - ref).abs().max()) <= 0.05 * scale


def test_squeeze_excite_tail_in_the_residual_launch_equals_the_separate_pass(monkeypatch):
    """ABI v7 (ts_tcs_desc.se_y / se_gate): internal stride-1 blocks close with relu(gate * main + residual) inside the residual 1x1 launch's
    epilogue instead of ts_se_apply_fwd over three tensors (citrinet/blocks.py:186-196).  Same arithmetic -- the residual rounded to bf16, one
    fma, ReLU, one rounding -- so a 6-block encoder (512 / 1024 output channels: both tile shapes of the split kernel, ragged lengths, a strided
    block in between that must keep the separate pass) gives identical bits either way; and the fused path is really taken."""
    import thunder_speech_amd.citrinet.blocks as cb
    filters, ks, strides = [512, 512, 1024, 1024], [11, 13, 15, 17], [1, 1, 2, 1]
    arch = otcs.citrinet_arch(filters=filters, kernel_sizes=ks, strides=strides, feat_in=64)
    sd = otcs.synth_encoder_state(arch, seed=7)
    enc = cb.CitrinetEncoder(filters=filters, kernel_sizes=ks, strides=strides, feat_in=64)
    enc.load_state_dict(sd, strict=True)
    enc = enc.cuda().eval()
    g = torch.Generator().manual_seed(3)
    x = bf16_round(torch.randn(3, 64, 700, generator=g)).cuda()
    lengths = torch.tensor([700, 433, 96]).cuda()
    calls = []
    real = cb.se_apply
    monkeypatch.setattr(cb, "se_apply", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    monkeypatch.setattr(cb, "FUSE_SE_TAIL", True)
    y1, l1 = enc(x, lengths)
    y1 = y1.clone()
    fused_calls = len(calls)
    monkeypatch.setattr(cb, "FUSE_SE_TAIL", False)
    y0, l0 = enc(x, lengths)
    n_se = sum(1 for m in enc.modules() if isinstance(m, cb.SqueezeExcite))
    assert len(calls) - fused_calls == n_se                   # the separate pass, once per squeeze-excite block
    assert 0 < fused_calls < n_se                            # fused: only strided / caller-visible blocks still take it
    assert torch.equal(l0, l1) and torch.equal(y0, y1)


def test_tiny_citrinet_matches_reference_fixture(golden):
    from thunder_speech_amd.citrinet.blocks import CitrinetEncoder
    g = golden("citrinet_tiny.npz")
    arch = otcs.citrinet_arch(filters=[32, 32], kernel_sizes=[5, 7], strides=[2, 1], feat_in=16)
    sd = otcs.synth_encoder_state(arch, seed=int(g["enc_seed"]))
    enc = CitrinetEncoder(filters=[32, 32], kernel_sizes=[5, 7], strides=[2, 1], feat_in=16)
    enc.load_state_dict(sd, strict=True)
    y, yl = enc.cuda().eval()(torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["lengths"]).cuda())
    assert np.array_equal(yl.cpu().numpy(), g["out_lengths"])
    got = y.float().cpu().numpy()[:, ::8, :]
    scale = max(1.0, float(np.abs(g["y_sample"]).max()))
    assert float(np.abs(got - g["y_sample"]).max()) <= 0.05 * scale      # bf16 storage through 4 blocks vs the fp32 reference
