"""Oracle conv stacks vs fixtures produced by the real reference blocks (CPU only)."""
import numpy as np
import pytest
import torch

from conftest import sd_from_npz
from oracle import frontend as fe
from oracle import primitives as prim
from oracle import tcs

BLOCK_CASES = {
    "qn_res_k11": dict(in_ch=16, out_ch=32, repeat=3, kernel=11),
    "qn_stride2": dict(in_ch=16, out_ch=24, repeat=1, kernel=33, stride=2, residual=False),
    "qn_dil2": dict(in_ch=24, out_ch=24, repeat=1, kernel=13, dilation=2, residual=False),
    "qn_dense_k1": dict(in_ch=24, out_ch=48, repeat=1, kernel=1, residual=False, separable=False),
    "qn_stride2_rep2": dict(in_ch=16, out_ch=16, repeat=2, kernel=5, stride=2, residual=True),
    "cn_s1": dict(in_ch=16, out_ch=32, repeat=3, kernel=7, stride=1, family="citrinet"),
    "cn_s2": dict(in_ch=32, out_ch=32, repeat=2, kernel=9, stride=2, family="citrinet"),
}


@pytest.mark.parametrize("name", sorted(BLOCK_CASES))
def test_block_eval_and_train_match_reference(golden, name):
    g = golden("blocks.npz")
    spec = tcs.BlockSpec(**BLOCK_CASES[name])
    sd = sd_from_npz(g, f"{name}/sd/")
    x, lengths = torch.from_numpy(g[f"{name}/x"]), torch.from_numpy(g[f"{name}/lengths"])
    y, out_len = tcs.block_forward(spec, sd, "", x, lengths)
    np.testing.assert_allclose(y.numpy(), g[f"{name}/y_eval"], atol=1e-4)      # reference atol (tests/utils.py:53)
    assert np.array_equal(out_len.numpy(), g[f"{name}/out_lengths"])
    stats = {}
    yt, _ = tcs.block_forward(spec, sd, "", x, lengths, training=True, new_stats=stats)
    np.testing.assert_allclose(yt.numpy(), g[f"{name}/y_train"], atol=1e-4)
    assert stats, "train mode must update running statistics"
    for k, v in stats.items():
        np.testing.assert_allclose(v.numpy(), g[f"{name}/new/" + k.replace(".", "/")], atol=1e-5)


@pytest.mark.parametrize("name", sorted(BLOCK_CASES))
def test_bf16_emulation_stays_close_to_fp32(golden, name):
    """The HIP-ordered (BN-folded, bf16-rounded) evaluation is the same function up to bf16 noise."""
    g = golden("blocks.npz")
    spec = tcs.BlockSpec(**BLOCK_CASES[name])
    sd = sd_from_npz(g, f"{name}/sd/")
    x, lengths = torch.from_numpy(g[f"{name}/x"]), torch.from_numpy(g[f"{name}/lengths"])
    y, _ = tcs.block_forward(spec, sd, "", prim.bf16_round(x), lengths, emulate_bf16=True)
    ref = g[f"{name}/y_eval"]
    assert np.abs(y.numpy() - ref).max() <= 0.03 * max(1.0, np.abs(ref).max())


def test_masked_conv_and_se(golden):
    g = golden("conv_se.npz")
    y, yl = tcs.masked_conv(torch.from_numpy(g["conv_x"]), torch.from_numpy(g["conv_lengths"]),
                            torch.from_numpy(g["conv_w"]), 1, 2, 1, 8)
    np.testing.assert_allclose(y.numpy(), g["conv_y"], atol=1e-6)
    assert np.array_equal(yl.numpy(), g["conv_out_lengths"]) and yl.dtype == torch.float32     # A5
    # frames >= length are zeroed before the conv (A2): changing them must not change the output
    x2 = torch.from_numpy(g["conv_x"]).clone()
    x2[1, :, 11:] = 123.0
    y2, _ = tcs.masked_conv(x2, torch.from_numpy(g["conv_lengths"]), torch.from_numpy(g["conv_w"]), 1, 2, 1, 8)
    assert torch.equal(y, y2)
    se = tcs.squeeze_excite(torch.from_numpy(g["se_x"]), torch.from_numpy(g["se_w1"]), torch.from_numpy(g["se_w2"]))
    np.testing.assert_allclose(se.numpy(), g["se_y"], atol=1e-6)


@pytest.mark.parametrize("k,s,d", [(11, 1, 1), (33, 2, 1), (87, 1, 2), (1, 1, 1), (5, 3, 1), (13, 1, 3)])
def test_same_padding_closed_form(k, s, d):
    # reference: tests/quartznet/test_blocks_qn.py:71-116 -- output length is ceil(L / stride)
    pad = prim.same_padding(k, s, d)
    for L in (17, 64, 101):
        out = prim.conv_out_length(torch.tensor([L]), k, s, pad, d)
        assert int(out) == (L + s - 1) // s or (k % 2 == 0)
    with pytest.raises(ValueError):
        prim.same_padding(3, 2, 2)


def test_state_dict_layout_counts():
    """225 entries for QN5x5, 635 for QN15x5 (126 / 356 trainable tensors)."""
    for rb, total, trainable in ((1, 225, 126), (3, 635, 356)):
        sd = tcs.synth_encoder_state(tcs.quartznet_arch(repeat_blocks=rb), seed=0)
        assert len(sd) == total
        assert sum(1 for k in sd if "running" not in k and "num_batches" not in k) == trainable
    sd = tcs.synth_encoder_state(tcs.quartznet_arch(repeat_blocks=3), seed=0)
    n = sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k)
    assert n + 1024 * 29 + 29 == 18924381
    assert sd["1.mconv.0.conv.weight"].shape == (256, 1, 33) and sd["1.mconv.1.conv.weight"].shape == (256, 256, 1)
    assert "1.res.0.conv.weight" in sd and "1.mconv.22.layer.0.running_var" in sd and "17.mconv.1.layer.0.bias" in sd


def test_quartznet5x5_end_to_end_logits(golden):
    g = golden("qn5x5_e2e.npz")
    rng = np.random.Generator(np.random.PCG64(int(g["wav_seed"])))
    wav = torch.from_numpy((0.1 * rng.standard_normal(tuple(g["wav_shape"]))).astype(np.float32))
    for b, z in enumerate(g["wav_zero_from"]):
        wav[b, int(z):] = 0
    arch = tcs.quartznet_arch(repeat_blocks=1)
    sd = tcs.synth_encoder_state(arch, seed=int(g["enc_seed"]), calibrate=True)
    dsd = tcs.synth_decoder_state(1024, 29, seed=int(g["dec_seed"]), gain=4.0)
    feats, fl = fe.filterbank_features(wav, torch.from_numpy(g["wav_lengths"]))
    enc, el = tcs.encoder_forward(arch, sd, feats, fl)
    logits = tcs.conv1d_decoder_forward(dsd, enc)
    assert np.array_equal(el.numpy(), g["out_lengths"])
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=2e-3)
    np.testing.assert_allclose(enc[:, ::64, ::10].numpy(), g["enc_sample"], atol=1e-3)
    from oracle import decode as dec
    labels = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]
    assert dec.decode_prediction(dec.argmax_classes(logits.numpy()), dec.Vocab(labels)) == list(g["strings"])


def test_tiny_citrinet_and_linear_decoder(golden):
    g = golden("citrinet_tiny.npz")
    arch = tcs.citrinet_arch(filters=[32, 32], kernel_sizes=[5, 7], strides=[2, 1], feat_in=16)
    sd = tcs.synth_encoder_state(arch, seed=int(g["enc_seed"]))
    y, yl = tcs.encoder_forward(arch, sd, torch.from_numpy(g["x"]), torch.from_numpy(g["lengths"]))
    assert np.array_equal(yl.numpy(), g["out_lengths"])
    np.testing.assert_allclose(y[:, ::8, :].numpy(), g["y_sample"], atol=1e-4)
    g2 = golden("linear_decoder.npz")
    out = tcs.linear_decoder_forward(sd_from_npz(g2, "sd/"), torch.from_numpy(g2["x"]))
    np.testing.assert_allclose(out.numpy(), g2["y"], atol=1e-5)
