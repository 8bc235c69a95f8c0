"""CPU checks of the round-2 oracle restatements (Philox, SpecAugment / SpecCutout, collate, audio prep, edit distance, the
training-mode block cases round 1 raised on) against the fixtures generated from the imported reference
(tests/golden/make_golden_r2.py), published known answers, and independent implementations."""
import math

import numpy as np
import torch

from conftest import sd_from_npz
from oracle import augment as oaug, dataprep as odp, metrics as omet, philox as ph, tcs as otcs


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32-10."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = ph.philox4x32_10(*[np.uint32(c) for c in ctr], *key)
        assert tuple(int(v) for v in got) == want


def test_philox_transforms_have_the_right_statistics():
    r = ph.philox(1234, ph.DITHER, np.arange(200000, dtype=np.uint64))
    u = ph.u01(r[0])
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 3e-3
    n0, n1 = ph.normal2(r[0], r[1])
    for n in (n0, n1):
        assert abs(n.mean()) < 6e-3 and abs(n.std() - 1.0) < 6e-3
    assert abs(np.corrcoef(n0, n1)[0, 1]) < 1e-2
    noise = ph.dither_noise(5, 3, 10001)
    assert noise.shape == (10001,) and abs(noise.std() - 1.0) < 0.03
    assert not np.array_equal(noise[:100], ph.dither_noise(5, 4, 100))          # another clip, another stream
    keep = ph.dropout_keep(77, 100003, 0.3)
    assert abs(keep.mean() - 0.7) < 5e-3


def test_spec_augment_oracle_reproduces_the_reference_masks(golden):
    g = golden("r2_misc.npz")
    x = torch.from_numpy(g["spec_x"])
    for i in range(3):
        n_time, tw, n_freq, fw = [int(v) for v in g[f"specaug{i}_cfg"]]
        torch.manual_seed(int(g[f"specaug{i}_seed"]))
        table = oaug.draw_table(oaug.torch_rand2, 64, 301, n_time=n_time, time_width=tw, n_freq=n_freq, freq_width=fw)
        assert np.array_equal(table, g[f"specaug{i}_table"])
        for f0, f1, t0, t1 in table.tolist():          # torchaudio's own assertion: mask_end - mask_start < mask_param
            assert (t1 - t0 < tw) if (f0, f1) == (0, 64) else (f1 - f0 < fw)
    assert torch.equal(oaug.apply_table(x, g["specaug0_table"]), torch.from_numpy(g["specaug0_y"]))
    for i in range(2):
        n, tw, fw = [int(v) for v in g[f"cutout{i}_cfg"]]
        torch.manual_seed(int(g[f"cutout{i}_seed"]))
        table = oaug.draw_table(oaug.torch_rand2, 64, 301, n_cutout=n, cut_time_width=tw, cut_freq_width=fw)
        assert np.array_equal(table, g[f"cutout{i}_table"])
        assert all(t1 - t0 < fw for _, _, t0, t1 in table.tolist())               # the time span uses freq_width (reference quirk)
    assert torch.equal(oaug.apply_table(x, g["cutout0_table"]), torch.from_numpy(g["cutout0_y"]))


def test_spec_augment_philox_draws_are_valid_spans():
    for seed in range(20):
        t = oaug.draw_table(oaug.philox_rand2(seed), 80, 1001, n_time=2, time_width=50, n_freq=2, freq_width=20, n_cutout=0)
        assert t.shape == (4, 4)
        for f0, f1, t0, t1 in t.tolist():
            assert 0 <= f0 <= f1 <= 80 and 0 <= t0 <= t1 <= 1001


def test_collate_oracle_matches_reference_fixture(golden):
    g = golden("r2_misc.npz")
    clips = [torch.from_numpy(g[f"collate_clip{i}"]) for i in range(5)]
    a, l, texts = odp.asr_collate([(c, f"text {i}") for i, c in enumerate(clips)])
    assert torch.equal(a, torch.from_numpy(g["collate_audio"])) and torch.equal(l, torch.from_numpy(g["collate_lengths"]))
    assert [int(t.split()[1]) for t in texts] == g["collate_order"].tolist() == [1, 0, 2, 3, 4]     # stable: the tie keeps input order


def test_resampler_restatement_against_an_independent_polyphase_filter():
    """torchaudio is absent: cross-check the restated sinc resampler on a band-limited signal against scipy's polyphase
    resampler (a different low-pass design, so agreement is to filter-design accuracy, away from the edges)."""
    from scipy.signal import resample_poly
    t = np.arange(44100) / 44100.0
    x = (0.5 * np.sin(2 * np.pi * 440 * t) + 0.3 * np.sin(2 * np.pi * 1234.5 * t + 0.3)).astype(np.float32)
    for new in (16000, 8000):
        y = odp.resample(torch.from_numpy(x)[None], 44100, new)[0].numpy()
        assert y.shape[0] == math.ceil(new * x.shape[0] / 44100)
        ref = resample_poly(x.astype(np.float64), new // math.gcd(new, 44100), 44100 // math.gcd(new, 44100))
        n = min(len(y), len(ref))
        assert np.abs(y[200:n - 200] - ref[200:n - 200]).max() < 5e-3
        tn = np.arange(len(y)) / new                   # and against the analytic signal itself
        want = 0.5 * np.sin(2 * np.pi * 440 * tn) + 0.3 * np.sin(2 * np.pi * 1234.5 * tn + 0.3)
        assert np.abs(y[200:-200] - want[200:-200]).max() < 5e-3
    up = odp.resample(torch.from_numpy(x[:8000])[None], 8000, 16000)[0]
    assert up.shape[0] == 16000
    k, width, orig, new = odp.sinc_resample_kernel(44100, 16000)
    assert (orig, new) == (441, 160) and k.shape == (160, 1, 2 * width + 441)


def test_preprocess_audio_restatement():
    g = torch.Generator().manual_seed(2)
    a = torch.randn(2, 5000, generator=g) + 0.25
    y = odp.preprocess_audio(a, 16000)
    assert y.shape == (1, 5000) and abs(float(y.mean())) < 1e-6
    assert torch.allclose(y, a.mean(0, keepdim=True) - a.mean(0, keepdim=True).mean(1), atol=1e-7)


def test_edit_distance_known_answers():
    assert omet.edit_distance("kitten", "sitting") == 3
    assert omet.edit_distance("", "abc") == 3 and omet.edit_distance("abc", "") == 3 and omet.edit_distance("", "") == 0
    assert omet.edit_distance("flaw", "lawn") == 2
    # torchmetrics' documented examples
    preds, target = ["this is the prediction", "there is an other sample"], ["this is the reference", "there is another one"]
    assert abs(omet.word_error_rate(preds, target) - 0.5) < 1e-12
    assert abs(omet.char_error_rate(preds, target) - 0.3415) < 1e-4


def _train_case(g, name, family):
    sd = sd_from_npz(g, f"{name}/sd/")
    spec = otcs.BlockSpec(16, 24, repeat=2, kernel=5, stride=2, family=family)
    x, lengths = torch.from_numpy(g[f"{name}/x"]), torch.from_numpy(g[f"{name}/lengths"])
    return spec, sd, x, lengths


def test_train_mode_strided_blocks_oracle_matches_reference_autograd(golden):
    """The oracle's train-mode evaluation of a strided Citrinet block (squeeze-excite, strided last repeat, strided residual)
    and a strided QuartzNet block (residual stride = stride ** repeat, A8): forward, running statistics and every gradient
    against the REAL reference's autograd."""
    g = golden("r2_misc.npz")
    for name, family in (("cn_train_s2", "citrinet"), ("qn_train_s2", "quartznet")):
        spec, sd, x, lengths = _train_case(g, name, family)
        sd_ref = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
        xr = x.clone().requires_grad_(True)
        new_stats = {}
        y, yl = otcs.block_forward(spec, sd_ref, "", xr, lengths, training=True, new_stats=new_stats)
        assert torch.equal(yl, torch.from_numpy(g[f"{name}/out_lengths"]))
        np.testing.assert_allclose(y.detach().numpy(), g[f"{name}/y"], atol=2e-5)
        (y * torch.from_numpy(g[f"{name}/w"])).sum().backward()
        np.testing.assert_allclose(xr.grad.numpy(), g[f"{name}/dx"], atol=5e-5)
        for k, v in sd_ref.items():
            if v.requires_grad:
                np.testing.assert_allclose(v.grad.numpy(), g[f"{name}/grad/" + k.replace(".", "/")], atol=2e-4, err_msg=k)
        after = sd_from_npz(g, f"{name}/sd_after/")
        for k, v in new_stats.items():
            np.testing.assert_allclose(v.numpy(), after[k].numpy(), atol=1e-5, err_msg=k)
