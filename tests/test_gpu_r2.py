"""GPU parity of the round-2 additions, all through the C ABI: Philox streams (bit-exact vs the oracle), dither inside the front
end, SpecAugment / SpecCutout inside the normaliser and standalone, dropout, audio prep + collate, edit distance / error rates,
device text encode, the training-mode block cases round 1 raised on (squeeze-excite, strided residual, dropout, linear decoder),
and the regression tests of the round-1 ADVICE findings."""
import math

import numpy as np
import pytest
import torch

from conftest import sd_from_npz
from oracle import augment as oaug, dataprep as odp, frontend as ofe, metrics as omet, philox as ph, tcs as otcs

pytestmark = pytest.mark.gpu
DEV = "cuda"


# ----------------------------------------------------------------------------------------------------------------- Philox
@pytest.mark.parametrize("n,p", [(4096, 0.25), (1003, 0.5), (5, 0.1), (70001, 0.9)])
def test_dropout_mask_is_the_oracles_philox_stream_bit_for_bit(n, p):
    from thunder_speech_amd import train_ops as T
    x = torch.randn(n, device=DEV) + 3.0                  # no zeros: the mask is readable from y
    y = T.Dropout.apply(x, p, 123456789)
    keep = torch.from_numpy(ph.dropout_keep(123456789, n, p)).to(DEV)
    assert torch.equal(y != 0, keep)
    torch.testing.assert_close(y[keep], x[keep] / (1.0 - p), rtol=1e-6, atol=0)
    # backward re-draws the same mask
    xg = x.clone().requires_grad_(True)
    T.Dropout.apply(xg, p, 123456789).sum().backward()
    torch.testing.assert_close(xg.grad, keep.float() / (1.0 - p), rtol=1e-6, atol=0)
    assert not torch.equal(T.Dropout.apply(x, p, 1) != 0, keep)     # another seed, another mask


def test_device_mask_draw_matches_the_oracle_philox_stream(monkeypatch):
    from thunder_speech_amd import rng
    from thunder_speech_amd.quartznet.spec_augment import SpecAugment, SpecCutout
    for seed in (1, 99, 2 ** 40 + 17):
        monkeypatch.setattr(rng, "next_seed", lambda s=seed: s)
        m = SpecAugment(freq_masks=3, time_masks=2, freq_width=27, time_width=100)
        m.rng = "philox"
        want = oaug.draw_table(oaug.philox_rand2(seed), 80, 2001, n_time=2, time_width=100, n_freq=3, freq_width=27)
        assert np.array_equal(m.draw(80, 2001, DEV).cpu().numpy(), want)
        c = SpecCutout(rect_masks=4, time_width=9, freq_width=33)
        c.rng = "philox"
        want = oaug.draw_table(oaug.philox_rand2(seed), 64, 1501, n_cutout=4, cut_time_width=9, cut_freq_width=33)
        assert np.array_equal(c.draw(64, 1501, DEV).cpu().numpy(), want)


# -------------------------------------------------------------------------------------------------------------- front end
def _wave(b=3, t=8000, seed=5):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((0.1 * g.standard_normal((b, t))).astype(np.float32))


def test_dither_in_the_front_end_is_the_oracles_noise(monkeypatch):
    """train mode: log-mel of (x + dither * oracle noise) through the oracle front end == the kernel's log-mel; a dither large
    enough to matter (the reference default 1e-5 is below fp32 resolution of the features)."""
    from thunder_speech_amd import rng
    from thunder_speech_amd.quartznet.transform import FilterbankFeatures
    monkeypatch.setattr(rng, "next_seed", lambda: 4242)
    x, lengths = _wave(), torch.tensor([8000.0, 6000.0, 4100.0])
    fb = FilterbankFeatures(dither=0.02).to(DEV).train()
    fb(x.to(DEV), lengths.to(DEV))
    got = fb.last_logmel().cpu()
    noise = torch.from_numpy(np.stack([ph.dither_noise(4242, b, x.shape[1]) for b in range(x.shape[0])]))
    st = ofe.filterbank_features(x + 0.02 * noise, lengths, return_stages=True)
    want = st["logmel"].transpose(1, 2)                             # [B, frames, mels]
    for b, n in enumerate(st["lengths"].tolist()):
        assert float((got[b, :n] - want[b, :n]).abs().max()) < 5e-3
    clean = ofe.filterbank_features(x, lengths, return_stages=True)["logmel"].transpose(1, 2)
    assert float((got[0] - clean[0]).abs().max()) > 0.05              # the dither really is in there
    fb.eval()
    fb(x.to(DEV), lengths.to(DEV))
    assert float((fb.last_logmel().cpu()[0] - clean[0]).abs().max()) < 2e-3     # eval: identity (transform.py:115)


def test_spec_augment_in_the_front_end_equals_eval_features_with_the_reference_masks_zeroed(golden):
    from thunder_speech_amd.quartznet.transform import FilterbankFeatures
    g = golden("r2_misc.npz")
    x, lengths = _wave(2, 48000), torch.tensor([48000.0, 31000.0])       # 301 frames, like the fixture
    plain = FilterbankFeatures().to(DEV).eval()
    base, bl = plain(x.to(DEV), lengths.to(DEV))
    base = base.float().cpu()
    n_time, tw, n_freq, fw = [int(v) for v in g["specaug0_cfg"]]
    fb = FilterbankFeatures(num_time_masks=n_time, num_freq_masks=n_freq, mask_time_width=tw, mask_freq_width=fw, dither=0.0).to(DEV).train()
    torch.manual_seed(int(g["specaug0_seed"]))
    y, yl = fb(x.to(DEV), lengths.to(DEV))
    assert torch.equal(yl, bl)
    want = oaug.apply_table(base, g["specaug0_table"])
    assert torch.equal(y.float().cpu(), want)
    assert int((want == 0).sum()) > int((base == 0).sum())
    n, tw, fw = [int(v) for v in g["cutout1_cfg"]]
    fb = FilterbankFeatures(num_cutout_masks=n, mask_time_width=tw, mask_freq_width=fw, dither=0.0).to(DEV).train()
    torch.manual_seed(int(g["cutout1_seed"]))
    y, _ = fb(x.to(DEV), lengths.to(DEV))
    assert torch.equal(y.float().cpu(), oaug.apply_table(base, g["cutout1_table"]))
    fb.eval()
    assert torch.equal(fb(x.to(DEV), lengths.to(DEV))[0].float().cpu(), base)


def test_standalone_spec_modules_match_the_reference_fixture(golden):
    from thunder_speech_amd.quartznet.spec_augment import SpecAugment, SpecCutout
    g = golden("r2_misc.npz")
    x = torch.from_numpy(g["spec_x"]).to(DEV)
    n_time, tw, n_freq, fw = [int(v) for v in g["specaug0_cfg"]]
    m = SpecAugment(freq_masks=n_freq, time_masks=n_time, freq_width=fw, time_width=tw).train()
    torch.manual_seed(int(g["specaug0_seed"]))
    y = m(x)
    assert torch.equal(y.cpu(), torch.from_numpy(g["specaug0_y"])) and torch.equal(x.cpu(), torch.from_numpy(g["spec_x"]))
    n, tw, fw = [int(v) for v in g["cutout0_cfg"]]
    c = SpecCutout(rect_masks=n, time_width=tw, freq_width=fw).train()
    torch.manual_seed(int(g["cutout0_seed"]))
    assert torch.equal(c(x).cpu(), torch.from_numpy(g["cutout0_y"]))
    assert c.eval()(x) is x


# ---------------------------------------------------------------------------------------------------- data prep / collate
def test_collate_matches_reference_fixture(golden):
    from thunder_speech_amd.data import asr_collate
    g = golden("r2_misc.npz")
    clips = [torch.from_numpy(g[f"collate_clip{i}"]) for i in range(5)]
    for on_gpu in (False, True):
        samples = [((c.to(DEV) if on_gpu else c), f"text {i}") for i, c in enumerate(clips)]
        a, l, texts = asr_collate(samples)
        assert a.is_cuda and torch.equal(a.cpu(), torch.from_numpy(g["collate_audio"]))
        assert l.dtype == torch.float32 and torch.equal(l.cpu(), torch.from_numpy(g["collate_lengths"]))
        assert [int(t.split()[1]) for t in texts] == g["collate_order"].tolist()
    a, l, _ = asr_collate([(torch.arange(7.0), "a"), (torch.ones(1, 1), "b"), (torch.arange(5.0).to(DEV), "c")])     # odd sizes: scalar path
    assert a.shape == (3, 7) and a[1].tolist() == [0, 1, 2, 3, 4, 0, 0] and a[2].tolist() == [1, 0, 0, 0, 0, 0, 0]


@pytest.mark.parametrize("channels,rate,t", [(2, 44100, 30011), (1, 8000, 9001), (1, 16000, 12345), (3, 22050, 7000), (1, 48000, 48000)])
def test_audio_prep_matches_the_oracle(channels, rate, t):
    from thunder_speech_amd.data import AudioFileLoader
    g = torch.Generator().manual_seed(rate + t)
    audio = 0.3 * torch.randn(channels, t, generator=g) + 0.1
    loader = AudioFileLoader(force_mono=True, sample_rate=16000)
    got = loader.preprocess_audio(audio, rate).cpu()
    want = odp.preprocess_audio(audio, rate, True, 16000)
    assert got.shape == want.shape == (1, math.ceil(16000 // math.gcd(16000, rate) * t / (rate // math.gcd(16000, rate))))
    assert float((got - want).abs().max()) < 2e-5
    # device input, and the result feeds asr_collate without leaving the GPU
    again = loader.preprocess_audio(audio.to(DEV), rate)
    assert again.is_cuda and torch.equal(again.cpu(), got)


# ------------------------------------------------------------------------------------------------------------- text side
def test_edit_distance_and_error_rates_match_the_oracle():
    from thunder_speech_amd.metrics import CharErrorRate, WordErrorRate, edit_distances
    rnd = np.random.Generator(np.random.PCG64(3))
    a = [rnd.integers(0, 6, rnd.integers(0, 300)).tolist() for _ in range(40)] + [[], [1, 2, 3], []]
    b = [rnd.integers(0, 6, rnd.integers(0, 300)).tolist() for _ in range(40)] + [[4, 4], [], []]
    got = edit_distances(a, b).cpu().tolist()
    assert got == [omet.edit_distance(x, y) for x, y in zip(a, b)]
    preds, target = ["this is the prediction", "there is an other sample", "", "same"], ["this is the reference", "there is another one", "x y", "same"]
    cer, wer = CharErrorRate(), WordErrorRate()
    assert abs(float(cer(preds, target)) - omet.char_error_rate(preds, target)) < 1e-6
    assert abs(float(wer(preds, target)) - omet.word_error_rate(preds, target)) < 1e-6
    cer(["abc"], ["abd"])                                                 # accumulates across updates like torchmetrics
    assert abs(float(cer.compute()) - omet.char_error_rate(preds + ["abc"], target + ["abd"])) < 1e-6
    cer.reset()
    assert float(cer.compute()) == 0.0


def test_device_text_encode_matches_reference_fixture(golden):
    from thunder_speech_amd.text_processing.transform import BatchTextTransformer
    g = golden("r2_misc.npz")
    labels = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]
    t1 = BatchTextTransformer(tokens=labels)
    e, l = t1.encode([str(s) for s in g["enc_texts"]], device=DEV)
    assert e.is_cuda and e.dtype == torch.int64 and np.array_equal(e.cpu().numpy(), g["enc1"]) and np.array_equal(l.cpu().numpy(), g["len1"])
    t2 = BatchTextTransformer(tokens=labels, start_token="<bos>", end_token="<eos>", unknown_token="<unk>")
    e, l = t2.encode([str(s) for s in g["enc_texts_unk"]], device=DEV)
    assert np.array_equal(e.cpu().numpy(), g["enc2"]) and np.array_equal(l.cpu().numpy(), g["len2"])
    e, l = t1.encode([str(s) for s in g["enc_texts_unk"]], device=DEV)
    assert np.array_equal(e.cpu().numpy(), g["enc3"]) and np.array_equal(l.cpu().numpy(), g["len3"])
    assert np.array_equal(t1.encode(["ab"], return_length=False, device=DEV).cpu().numpy(), [[1, 2]])


def test_validation_step_reports_error_rates():
    from thunder_speech_amd.module import greedy_decode
    from thunder_speech_amd.registry import load_pretrained
    m = load_pretrained("QuartzNet5x5_synthetic").to(DEV).eval()
    wav, lengths = _wave(2, 16000).to(DEV), torch.tensor([16000.0, 12000.0], device=DEV)
    texts = ["hello world", "abc"]
    with torch.no_grad():
        loss = m.validation_step((wav, lengths, texts), 0)
        _, collapsed, counts = greedy_decode(m(wav, lengths)[0])
    assert torch.isfinite(loss)
    preds = m.text_transform.decode_collapsed(collapsed, counts)
    assert abs(float(m.validation_cer.compute()) - omet.char_error_rate(preds, texts)) < 1e-6
    assert abs(float(m.validation_wer.compute()) - omet.word_error_rate(preds, texts)) < 1e-6


# ------------------------------------------------------------------------------------------- training-mode block coverage
def _block(cls, sd, **kw):
    blk = cls(16, 24, repeat=2, kernel_size=(5,), stride=(2,), separable=True, **kw)
    blk.load_state_dict(sd, strict=True)
    return blk.to(DEV).train()


@pytest.mark.parametrize("name", ["cn_train_s2", "qn_train_s2"])
def test_train_mode_se_and_strided_residual_match_the_reference_autograd(golden, name):
    """Forward, running statistics, dL/dx and every parameter gradient against the REAL reference blocks' autograd."""
    from thunder_speech_amd.citrinet.blocks import CitrinetBlock
    from thunder_speech_amd.quartznet.blocks import QuartznetBlock
    g = golden("r2_misc.npz")
    sd = sd_from_npz(g, f"{name}/sd/")
    blk = _block(CitrinetBlock if name.startswith("cn") else QuartznetBlock, sd)
    x = torch.from_numpy(g[f"{name}/x"]).to(DEV).requires_grad_(True)
    y, yl = blk(x, torch.from_numpy(g[f"{name}/lengths"]).to(DEV))
    assert np.array_equal(yl.cpu().numpy(), g[f"{name}/out_lengths"])
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"{name}/y"], atol=2e-4)
    (y * torch.from_numpy(g[f"{name}/w"]).to(DEV)).sum().backward()
    want = g[f"{name}/dx"]
    assert float(np.abs(x.grad.cpu().numpy() - want).max()) <= 2e-3 * max(float(np.abs(want).max()), 1e-3)
    for k, p in blk.named_parameters():
        want = g[f"{name}/grad/" + k.replace(".", "/")]
        assert float(np.abs(p.grad.cpu().numpy() - want).max()) <= 2e-3 * max(float(np.abs(want).max()), 1e-3), k
    after = sd_from_npz(g, f"{name}/sd_after/")
    for k, v in blk.state_dict().items():
        if "running_" in k:
            np.testing.assert_allclose(v.cpu().numpy(), after[k].numpy(), atol=2e-5, err_msg=k)


def test_train_mode_dropout_in_blocks(golden, monkeypatch):
    """dropout > 0 (quartznet/blocks.py:227-228): the block's output is the p = 0 computation with the oracle's Philox masks
    applied after each ReLU -- checked through linearity of the LAST dropout (mout) and through determinism by seed."""
    from thunder_speech_amd import rng
    from thunder_speech_amd.quartznet.blocks import QuartznetBlock
    g = golden("r2_misc.npz")
    sd = sd_from_npz(g, "qn_train_s2/sd/")
    x = torch.from_numpy(g["qn_train_s2/x"]).to(DEV)
    lengths = torch.from_numpy(g["qn_train_s2/lengths"]).to(DEV)
    seeds = iter(range(1000, 2000))
    monkeypatch.setattr(rng, "next_seed", lambda: next(seeds))
    blk = _block(QuartznetBlock, sd, dropout=0.3)
    y1, _ = blk(x, lengths)
    seeds = iter(range(1000, 2000))
    blk2 = _block(QuartznetBlock, sd, dropout=0.3)
    y2, _ = blk2(x, lengths)
    assert torch.equal(y1, y2)                                           # same seeds, same masks
    frac = float((y1 == 0).float().mean())
    blk0 = _block(QuartznetBlock, sd, dropout=0.0)
    y0, _ = blk0(x, lengths)
    z0 = float((y0 == 0).float().mean())
    assert frac > z0 + 0.2 * (1.0 - z0)                                  # the final dropout alone zeroes ~30 % of what ReLU left
    # only the final dropout active: y = y0 * keep / (1 - p) with the oracle's mask for the first seed drawn
    blk3 = _block(QuartznetBlock, sd, dropout=0.0)
    blk3.mout[1].layer[0].p = 0.5
    seeds = iter([777])
    y3, _ = blk3(x, lengths)
    keep = torch.from_numpy(ph.dropout_keep(777, y0.numel(), 0.5)).to(DEV).view_as(y0)
    torch.testing.assert_close(y3, y0 * keep / 0.5, rtol=1e-6, atol=1e-7)
    blk3.eval()                                                          # eval: dropout is the identity (and the fused path runs)
    assert torch.isfinite(blk3(x, lengths)[0]).all()


def test_linear_decoder_trains(monkeypatch):
    """linear_decoder (blocks.py:226-248) in train mode: logits and gradients vs torch autograd; dropout = the oracle's mask."""
    from thunder_speech_amd import rng
    from thunder_speech_amd.blocks import linear_decoder
    torch.manual_seed(0)
    dec = linear_decoder(48, 11, 0.0).to(DEV).train()
    x = torch.randn(3, 48, 37, device=DEV, requires_grad=True)
    y = dec(x)
    ref = torch.nn.functional.linear(x.detach().cpu().transpose(1, 2).double(), dec[2].weight.detach().cpu().double(),
                                     dec[2].bias.detach().cpu().double()).transpose(1, 2)
    assert float((y.detach().cpu().double() - ref).abs().max()) < 1e-4
    cot = torch.randn_like(y)
    (y * cot).sum().backward()
    xr = x.detach().cpu().double().requires_grad_(True)
    w = dec[2].weight.detach().cpu().double().requires_grad_(True)
    b = dec[2].bias.detach().cpu().double().requires_grad_(True)
    (torch.nn.functional.linear(xr.transpose(1, 2), w, b).transpose(1, 2) * cot.cpu().double()).sum().backward()
    for got, want in ((x.grad, xr.grad), (dec[2].weight.grad, w.grad), (dec[2].bias.grad, b.grad)):
        assert float((got.cpu().double() - want).abs().max()) <= 1e-4 * max(float(want.abs().max()), 1.0)
    # frozen encoder (x without grad): the head alone trains
    dec.zero_grad()
    dec(x.detach()).sum().backward()
    assert dec[2].weight.grad is not None and float(dec[2].weight.grad.abs().sum()) > 0
    # dropout
    monkeypatch.setattr(rng, "next_seed", lambda: 31337)
    dec2 = linear_decoder(48, 11, 0.25).to(DEV).train()
    dec2.load_state_dict(dec.state_dict())
    keep = torch.from_numpy(ph.dropout_keep(31337, x.numel(), 0.25)).to(DEV).view_as(x)
    want = dec((x.detach() * keep / 0.75))
    torch.testing.assert_close(dec2(x.detach()), want, rtol=1e-5, atol=1e-5)
    dec2.eval()
    with torch.no_grad():
        assert float((dec2(x.detach()) - dec(x.detach())).abs().max()) < 0.05          # eval: fused bf16 kernel, no dropout


# ------------------------------------------------------------------------------------------------ ADVICE regression tests
def test_eval_forward_sees_fused_optimizer_and_running_stat_updates():
    """ADVICE (high): eval forward, FusedAdamW step in train mode, eval forward again -- the second eval must use the new
    weights AND the new BatchNorm running statistics, i.e. equal a fresh module loaded from the state_dict."""
    from thunder_speech_amd.optim import FusedAdamW
    from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
    arch = otcs.quartznet_arch(repeat_blocks=1)
    kw = dict(encoder_state=otcs.synth_encoder_state(arch, seed=0, calibrate=True), decoder_state=otcs.synth_decoder_state(1024, 29, seed=1))
    m = build_synthetic_quartznet(repeat_blocks=1, **kw).to(DEV).eval()
    wav = _wave(2, 16000).to(DEV)
    lengths = torch.tensor([16000.0, 12000.0], device=DEV)
    with torch.no_grad():
        before = m(wav, lengths)[0].clone()
    m.train()
    opt = FusedAdamW(m.parameters(), lr=5e-3, weight_decay=0.0)
    for _ in range(2):
        opt.zero_grad()
        m.training_step((wav, lengths, ["hello", "abc"]), 0).backward()
        opt.step()
    m.eval()
    with torch.no_grad():
        after = m(wav, lengths)[0].clone()
    fresh = build_synthetic_quartznet(repeat_blocks=1).to(DEV).eval()
    fresh.load_state_dict(m.state_dict())
    with torch.no_grad():
        want = fresh(wav, lengths)[0]
    assert float((after - before).abs().max()) > 1e-2                    # the update is visible ...
    assert torch.equal(after, want)                                      # ... and it is exactly the stored weights' output


def test_graphed_forward_tracks_new_lengths():
    """ADVICE (medium): replaying the front end + frozen encoder from a hipGraph with different FLOAT lengths per call."""
    from thunder_speech_amd.registry import load_pretrained
    m = load_pretrained("QuartzNet5x5_synthetic").to(DEV).eval()
    for p in m.encoder.parameters():
        p.requires_grad = False
    wav = _wave(3, 16000).to(DEV)
    l1 = torch.tensor([16000.0, 16000.0, 16000.0], device=DEV)
    l2 = torch.tensor([16000.0, 9000.0, 5000.0], device=DEV)
    with torch.no_grad():
        e1, e2 = m(wav, l1)[0].clone(), m(wav, l2)[0].clone()
    m.graph_frozen_encoder()
    with torch.no_grad():
        g1 = m(wav, l1)[0].clone()
        g2 = m(wav, l2)[0].clone()
        g1b = m(wav, l1)[0].clone()
    assert torch.equal(g1, e1) and torch.equal(g2, e2) and torch.equal(g1b, e1)
    assert not torch.equal(e1, e2)


def test_frozen_decoder_with_trainable_encoder_keeps_the_autograd_path():
    """ADVICE (medium): decoder parameters frozen, encoder trainable -- logits must come from the fp32 training ops and the
    gradient must reach the encoder."""
    from thunder_speech_amd.registry import load_pretrained
    m = load_pretrained("QuartzNet5x5_synthetic").to(DEV).train()
    for p in m.decoder.parameters():
        p.requires_grad = False
    wav = _wave(2, 16000).to(DEV)
    loss = m.training_step((wav, torch.tensor([16000.0, 16000.0], device=DEV), ["hello", "abc"]), 0)
    loss.backward()
    assert torch.isfinite(loss)
    grads = [p.grad for p in m.encoder.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in grads) and sum(float(g.abs().sum()) for g in grads) > 0
    assert all(p.grad is None for p in m.decoder.parameters())


def test_ctc_concatenated_targets_and_wild_ids(golden):
    """ADVICE (low): 1-D targets are F.ctc_loss's concatenated format; ids outside [0, V) make the utterance infeasible
    (loss contribution 0, gradient 0) instead of indexing out of bounds."""
    from thunder_speech_amd.ctc_loss import calculate_ctc
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(3, 7, 40, generator=g)
    tl = torch.tensor([5, 3, 4])
    padded = torch.zeros(3, 5, dtype=torch.long)
    cat = []
    for i, n in enumerate(tl.tolist()):
        row = torch.randint(0, 6, (n,), generator=g)
        padded[i, :n] = row
        cat.append(row)
    il = torch.tensor([40, 33, 21])
    ref = torch.nn.functional.ctc_loss(logits.permute(2, 0, 1).log_softmax(2), padded, il, tl, blank=6, reduction="mean", zero_infinity=True)
    a = calculate_ctc(logits.to(DEV), padded.to(DEV), il.to(DEV), tl.to(DEV), 6)
    b = calculate_ctc(logits.to(DEV), torch.cat(cat).to(DEV), il.to(DEV), tl.to(DEV), 6)
    assert abs(float(a) - float(ref)) < 1e-4 and abs(float(b) - float(ref)) < 1e-4
    wild = padded.clone()
    wild[1, 1] = 99
    lg = logits.to(DEV).requires_grad_(True)
    c = calculate_ctc(lg, wild.to(DEV), il.to(DEV), tl.to(DEV), 6)
    c.backward()
    keep = [0, 2]
    ref2 = torch.nn.functional.ctc_loss(logits[keep].permute(2, 0, 1).log_softmax(2), padded[keep], il[keep], tl[keep], blank=6,
                                        reduction="sum", zero_infinity=True)
    per = torch.nn.functional.ctc_loss(logits[keep].permute(2, 0, 1).log_softmax(2), padded[keep], il[keep], tl[keep], blank=6,
                                       reduction="none", zero_infinity=True)
    want = float((per / tl[keep]).sum() / 3)
    assert abs(float(c) - want) < 1e-4 and float(lg.grad[1].abs().max()) == 0.0 and torch.isfinite(lg.grad).all()


def test_gradient_sync_on_the_gpu_flat_views_and_wire_format():
    """parallel.GradientSync on one GPU: gradients accumulate into the flat buffer through the .grad views, FusedAdamW consumes the
    views, and the bf16 wire kernels round-trip to bf16(grad / world)."""
    from thunder_speech_amd import _lib
    from thunder_speech_amd.optim import FusedAdamW
    from thunder_speech_amd.parallel import GradientSync
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in ((29, 64, 1), (29,), (5, 7), (1003,), (256, 256, 1), (256, 1, 33), (256,), (256,), (512, 256, 1))]
    sync = GradientSync(ps, bucket_bytes=8192)
    assert len(sync.buckets) >= 3 and sync.flat.is_cuda
    opt, ref = FusedAdamW(ps, lr=1e-2), None
    ps_ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt_ref = torch.optim.AdamW(ps_ref, lr=1e-2)
    for step in range(3):
        sync.zero_grad()
        sum((p * p).sum() * (i + 1) for i, p in enumerate(ps)).backward()
        sync.finish()
        for i, p in enumerate(ps):
            assert p.grad.data_ptr() >= sync.flat.data_ptr() and torch.allclose(p.grad, 2 * (i + 1) * p.detach())
        opt.step()
        opt_ref.zero_grad()
        sum((p * p).sum() * (i + 1) for i, p in enumerate(ps_ref)).backward()
        opt_ref.step()
    for p, q in zip(ps, ps_ref):
        torch.testing.assert_close(p, q, rtol=2e-5, atol=2e-6)
    g = torch.randn(100003, device=DEV)[:100000]                      # 16-byte aligned, length not a multiple of 8 below
    for n in (100000, 99997, 5):
        wire = torch.empty(n, dtype=torch.bfloat16, device=DEV)
        back = torch.empty(n, dtype=torch.float32, device=DEV)
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().ts_grad_wire_pack(g.data_ptr(), wire.data_ptr(), n, 0.125, st), "pack")
        _lib.check(_lib.lib().ts_grad_wire_unpack(wire.data_ptr(), back.data_ptr(), n, st), "unpack")
        assert torch.equal(wire, (g[:n] * 0.125).to(torch.bfloat16)) and torch.equal(back, wire.float())
