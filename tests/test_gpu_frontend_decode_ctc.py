"""GPU parity of the mel front end, greedy decode and CTC kernels (through the C ABI) vs oracle + fixtures."""
import numpy as np
import pytest
import torch

from oracle import ctc as octc
from oracle import decode as odec
from oracle import frontend as ofe

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,kw", [("qn", {}), ("cn", dict(n_window_size=400, nfilt=80))])
def test_frontend_matches_reference_fixture(golden, tag, kw):
    from thunder_speech_amd.quartznet.transform import FilterbankFeatures
    g = golden(f"frontend_{tag}.npz")
    fb = FilterbankFeatures(**kw).cuda().eval()
    feats, flen = fb(torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["lengths"]).cuda())
    torch.cuda.synchronize()
    assert feats.shape == g["features"].shape
    assert np.array_equal(flen.cpu().numpy(), g["feat_lengths"]) and flen.dtype == torch.int64
    logmel = fb.last_logmel().cpu().numpy().transpose(0, 2, 1)           # [B, mel, frame]
    np.testing.assert_allclose(logmel, g["logmel"], atol=2e-3)           # fp32 FFT vs torch.stft
    got = feats.float().cpu().numpy()
    # output is bf16: |x| <= ~4 -> half-ulp 2^-7 ; the reference's own CPU/GPU tolerance is 1e-3 on top
    np.testing.assert_allclose(got, g["features"], atol=2e-2)
    for b, n in enumerate(g["feat_lengths"]):
        assert np.all(got[b, :, n:] == 0)


def test_frontend_long_clip_matches_oracle():
    from thunder_speech_amd.quartznet.transform import FilterbankFeatures
    rng = np.random.Generator(np.random.PCG64(5))
    x = torch.from_numpy((0.1 * rng.standard_normal((3, 48000))).astype(np.float32))
    x[1, 40000:] = 0
    lengths = torch.tensor([48000, 40000, 47999])
    ref = ofe.filterbank_features(x, lengths, return_stages=True)
    fb = FilterbankFeatures().cuda().eval()
    feats, flen = fb(x.cuda(), lengths.cuda())
    assert torch.equal(flen.cpu(), ref["lengths"])
    np.testing.assert_allclose(fb.last_logmel().cpu().numpy().transpose(0, 2, 1), ref["logmel"].numpy(), atol=2e-3)
    np.testing.assert_allclose(feats.float().cpu().numpy(), ref["features"].numpy(), atol=2e-2)


def test_greedy_decode_matches_oracle_and_reference_cases():
    from thunder_speech_amd.module import greedy_decode
    from thunder_speech_amd.text_processing.transform import BatchTextTransformer
    labels = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]
    tt = BatchTextTransformer(tokens=list(labels))
    vocab = odec.Vocab(list(labels))
    rng = np.random.Generator(np.random.PCG64(9))
    for t in (1, 7, 256, 257, 751):
        logits = rng.standard_normal((5, 29, t)).astype(np.float32)
        logits[1, 28, :] += 10.0                               # all blank
        logits[2, 3, : t // 2] += 10.0                         # long run then noise
        ids, collapsed, counts = greedy_decode(torch.from_numpy(logits).cuda())
        ref_ids = odec.argmax_classes(logits)
        assert np.array_equal(ids.cpu().numpy(), ref_ids)
        strings = tt.decode_collapsed(collapsed, counts)
        assert strings == odec.decode_prediction(ref_ids, vocab)
        assert strings == tt.decode_prediction(ids)            # host path of the same API
        for b in range(5):
            n = int(counts[b])
            assert np.array_equal(collapsed[b, :n].cpu().numpy(), odec.collapse_repeats(ref_ids[b]))
    # reference known answers (tests/text/test_transforms.py:59-91)
    a, bb, blank = vocab.stoi["a"], vocab.stoi["b"], vocab.blank_idx

    def one_hot(seq):
        lg = np.zeros((1, 29, len(seq)), dtype=np.float32)
        lg[0, seq, np.arange(len(seq))] = 5.0
        return torch.from_numpy(lg).cuda()
    for seq, want in (([blank] * 10, ""), ([a] * 5 + [bb] * 5, "ab"), ([a] * 4 + [blank] + [a] * 4, "aa")):
        _, col, cnt = greedy_decode(one_hot(seq))
        assert tt.decode_collapsed(col, cnt) == [want]


def test_ctc_loss_and_grad_match_reference_fixture(golden):
    from thunder_speech_amd.ctc_loss import calculate_ctc
    g = golden("ctc.npz")
    logits = torch.from_numpy(g["logits"]).cuda().requires_grad_(True)
    loss = calculate_ctc(logits, torch.from_numpy(g["targets"]).cuda(), torch.from_numpy(g["input_lengths"]).cuda(),
                         torch.from_numpy(g["target_lengths"]).cuda(), int(g["blank"]))
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logits.grad.cpu().numpy(), g["grad"], atol=2e-5)


def test_ctc_matches_oracle_on_training_like_shapes():
    from thunder_speech_amd.ctc_loss import calculate_ctc
    rng = np.random.Generator(np.random.PCG64(4))
    B, V, T = 6, 29, 251
    logits = (1.5 * rng.standard_normal((B, V, T))).astype(np.float32)
    tl = rng.integers(30, 100, B)
    tg = np.full((B, 100), 28, dtype=np.int64)
    for b in range(B):
        tg[b, : tl[b]] = rng.integers(0, 28, tl[b])
    il = rng.integers(200, T + 1, B).astype(np.float32)
    want_loss, want_grad, _ = octc.calculate_ctc(logits, tg, il, tl, 28)
    lg = torch.from_numpy(logits).cuda().requires_grad_(True)
    loss = calculate_ctc(lg, torch.from_numpy(tg).cuda(), torch.from_numpy(il).cuda(), torch.from_numpy(tl).cuda(), 28)
    (2.0 * loss).backward()
    np.testing.assert_allclose(float(loss.detach()), want_loss, rtol=2e-5)
    np.testing.assert_allclose(lg.grad.cpu().numpy(), 2.0 * want_grad, atol=2e-5)


@pytest.mark.parametrize("tg_dtype,len_dtype", [(torch.int64, torch.int64), (torch.int32, torch.float32), (torch.int64, torch.float64), (torch.int16, torch.int32)])
def test_ctc_prepare_kernel_equals_the_torch_rules(tg_dtype, len_dtype):
    """ts_ctc_prepare (one launch) against ctc_loss._prepare_targets + the input-length rules it replaces for device targets:
    padding and wild ids rewritten to 0, an utterance with a wild id below its length made infeasible, `.long()` truncation."""
    from thunder_speech_amd import ctc_loss
    g = torch.Generator().manual_seed(11)
    b, s, v = 9, 37, 29
    tg = torch.randint(0, v, (b, s), generator=g)
    tl = torch.randint(0, s + 1, (b,), generator=g)
    tl[0], tl[1] = 0, s
    tg[2, int(tl[2]) :] = 999                      # wild ids in the padding only: harmless
    tg[3, 0] = -1                                  # wild ids below the length: rows 3 and 4 become infeasible
    tl[3] = max(int(tl[3]), 1)
    tg[4, 5] = v
    tl[4] = max(int(tl[4]), 6)
    il = torch.randint(40, 200, (b,), generator=g).to(torch.float64) + 0.75
    tgd, tld, ild = tg.to(tg_dtype).cuda(), tl.to(len_dtype).cuda(), il.to(len_dtype).cuda()
    want_tg, want_tl, bad = ctc_loss._prepare_targets(tgd, tld, b, v, tgd.device)
    want_il = ild.to(torch.int64).to(torch.int32)
    want_il = torch.where(bad, torch.zeros_like(want_il), want_il)
    want_tl = torch.where(bad & (want_tl == 0), torch.ones_like(want_tl), want_tl)
    dropped = ctc_loss.bad_target_rows(tgd.device)
    got_tg, got_tl, got_il = ctc_loss._prepare_on_device(tgd, tld, ild, b, v, tgd.device)
    assert ctc_loss.bad_target_rows(tgd.device) == dropped + 2          # the device-side count of dropped utterances (ADVICE: no silent shrinking)
    assert bad.cpu().tolist() == [False, False, False, True, True] + [False] * 4
    assert torch.equal(got_tg, want_tg) and torch.equal(got_tl, want_tl) and torch.equal(got_il, want_il)
    # no labels at all: [B, 0] targets behave like one padded column
    e_tg, e_tl, e_il = ctc_loss._prepare_on_device(tgd[:, :0], torch.zeros_like(tld), ild, b, v, tgd.device)
    assert e_tg.shape == (b, 1) and int(e_tg.abs().sum()) == 0 and int(e_tl.abs().sum()) == 0 and torch.equal(e_il, ild.to(torch.int64).to(torch.int32))


def test_ctc_long_transcripts_take_two_states_per_thread():
    """More than 511 labels: 2S+1 > 1024 extended states, i.e. the recursion kernel's two-states-per-thread instantiation; ragged
    lengths, one infeasible utterance (fewer frames than labels)."""
    from thunder_speech_amd.ctc_loss import calculate_ctc
    rng = np.random.Generator(np.random.PCG64(5))
    B, V, T, S = 3, 12, 1500, 700
    logits = rng.standard_normal((B, V, T)).astype(np.float32)
    tl = np.array([700, 520, 650])
    tg = np.zeros((B, S), dtype=np.int64)
    for b in range(B):
        tg[b, : tl[b]] = rng.integers(0, V - 1, tl[b])
    il = np.array([1500, 1377, 600])               # the last one cannot emit 650 labels in 600 frames: loss 0, gradient 0
    want_loss, want_grad, _ = octc.calculate_ctc(logits, tg, il, tl, V - 1)
    lg = torch.from_numpy(logits).cuda().requires_grad_(True)
    loss = calculate_ctc(lg, torch.from_numpy(tg).cuda(), torch.from_numpy(il).cuda(), torch.from_numpy(tl).cuda(), V - 1)
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), want_loss, rtol=5e-5)
    np.testing.assert_allclose(lg.grad.cpu().numpy(), want_grad, atol=2e-5)
    assert float(lg.grad[2].abs().max()) == 0.0


def test_error_rate_call_returns_the_batch_value_and_compute_the_running_one():
    """torchmetrics' forward semantics: metric(preds, target) is THIS batch's value, compute() the accumulated one."""
    from thunder_speech_amd.metrics import CharErrorRate, WordErrorRate
    cer = CharErrorRate()
    assert abs(float(cer(["abc"], ["abd"])) - 1 / 3) < 1e-6
    assert float(cer(["xyz"], ["xyz"])) == 0.0
    assert abs(float(cer.compute()) - 1 / 6) < 1e-6
    wer = WordErrorRate()
    assert abs(float(wer(["a b c d"], ["a x c d"])) - 0.25) < 1e-6 and abs(float(wer(["q"], ["q r"])) - 0.5) < 1e-6
    assert abs(float(wer.compute()) - 2 / 6) < 1e-6


@pytest.mark.parametrize("T", [20000, 45000])
def test_ctc_very_long_clip_uses_the_large_lds_configuration(T):
    """20 000 frames (a 6.7-minute clip after the stride-2 stem): the per-frame log-sum-exp row no longer fits the default 64 KiB of LDS.
    45 000 frames (15 minutes): it does not fit the 160 KiB either and is read back from its global copy (prefetched with the emissions).
    The blank is made likely, as in a trained model, so that the path scores stay where f32 log-domain arithmetic resolves them (with flat
    random logits a 20 000-frame score is ~ -42 000, where one f32 ulp is 0.004 -- any f32 CTC is then only good to a percent)."""
    from thunder_speech_amd.ctc_loss import calculate_ctc
    rng = np.random.Generator(np.random.PCG64(6))
    B, V = 2, 8
    logits = rng.standard_normal((B, V, T)).astype(np.float32)
    logits[:, V - 1] += 7.0
    tl = np.array([40, 25])
    tg = np.zeros((B, 40), dtype=np.int64)
    for b in range(B):
        tg[b, : tl[b]] = rng.integers(0, V - 1, tl[b])
    il = np.array([T, 12345])
    lg = torch.from_numpy(logits).cuda().requires_grad_(True)
    loss = calculate_ctc(lg, torch.from_numpy(tg).cuda(), torch.from_numpy(il).cuda(), torch.from_numpy(tl).cuda(), V - 1)
    loss.backward()
    ref_in = torch.from_numpy(logits).double().requires_grad_(True)
    ref = torch.nn.functional.ctc_loss(ref_in.permute(2, 0, 1).log_softmax(2), torch.from_numpy(tg), torch.from_numpy(il), torch.from_numpy(tl), blank=V - 1,
                                       reduction="mean", zero_infinity=True)
    ref.backward()
    np.testing.assert_allclose(float(loss.detach()), float(ref), rtol=2e-5)
    want = ref_in.grad.numpy()
    # f32 log-domain sums over T steps: the rounding of the path scores grows with the clip (1 % of the largest entry at 20 000 frames)
    np.testing.assert_allclose(lg.grad.cpu().numpy(), want, atol=1e-2 * (T / 20000) * float(np.abs(want).max()))


@pytest.mark.parametrize("v,t", [(1024, 251), (1024, 64), (257, 700), (1024, 1025), (3, 5), (1000, 1)])
def test_greedy_decode_large_vocabularies_and_ties(v, t):
    """Clips shorter than the decode workgroup split the CLASSES over its threads (Citrinet: 251 frames x 1 024 sentencepiece classes);
    quantised logits force ties, which must resolve to the lowest index like torch.argmax, within and across class slices."""
    from thunder_speech_amd.module import greedy_decode
    rng = np.random.Generator(np.random.PCG64(v + t))
    logits = np.round(rng.standard_normal((3, v, t)) * 2.0).astype(np.float32) / 2.0      # many exact ties
    logits[1, :, : max(t // 3, 1)] = 0.25                                                   # every class ties: index 0 wins
    ids, collapsed, counts = greedy_decode(torch.from_numpy(logits).cuda())
    ref_ids = odec.argmax_classes(logits)
    assert np.array_equal(ref_ids, logits.argmax(1))                                         # the oracle's rule is numpy's / torch's: first maximum
    assert np.array_equal(ids.cpu().numpy(), ref_ids)
    for b in range(3):
        assert np.array_equal(collapsed[b, : int(counts[b])].cpu().numpy(), odec.collapse_repeats(ref_ids[b]))


def test_convolution_stft_and_fourier_matrix_match_the_reference_tests():
    """The reference's own checks (tests/test_blocks.py:8-30): the Fourier matrix equals fft(eye) to 1e-3, convolution_stft equals
    torch.stft(return_complex=False) to 1e-2 -- plus the oracle's float64 STFT at 1e-4 on a second geometry."""
    from thunder_speech_amd.blocks import _fourier_matrix, convolution_stft
    for n in (64, 128, 256, 512, 1024):
        assert torch.allclose(torch.fft.fft(torch.eye(n)), _fourier_matrix(n, "cpu"), atol=1e-3)
    torch.manual_seed(0)
    x = torch.randn(10, 1000)
    window = torch.hann_window(256, periodic=False)
    got = convolution_stft(x.cuda(), n_fft=1024, hop_length=512, win_length=256, window=window).cpu()
    want = torch.view_as_real(torch.stft(x, n_fft=1024, hop_length=512, win_length=256, window=window, return_complex=True))
    assert got.shape == want.shape == (10, 513, 2, 2) and torch.allclose(got, want, atol=1e-2)
    x2 = torch.randn(3, 4000, dtype=torch.float64)
    w2 = torch.hann_window(300, periodic=False, dtype=torch.float64)
    got2 = convolution_stft(x2.float().cuda(), n_fft=400, hop_length=160, win_length=300, window=w2.float()).cpu()
    want2 = torch.view_as_real(torch.stft(x2, n_fft=400, hop_length=160, win_length=300, window=w2, return_complex=True)).float()
    assert got2.shape == want2.shape and float((got2 - want2).abs().max()) <= 1e-4 * float(want2.abs().max())
