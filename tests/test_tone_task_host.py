"""CPU tests of the synthetic transcription task behind the trained-weights transcript check (tools/train_margin_model.py): the frame-label
patterns, their CTC reading and the audio synthesis -- the properties the GPU test and bench.py's `check_trained` rely on."""
import numpy as np
import torch

from tools.train_margin_model import BURST_FRAMES, SLOT_FRAMES, TONE_HZ, encoder_frames, frame_labels, tone_clips, transcript


def test_encoder_frames_follow_the_reference_length_rules():
    # mel frames = n // 160 + 1 (quartznet/transform.py:182-184), halved by the stride-2 stem conv (quartznet/blocks.py:149-155)
    assert [encoder_frames(16000 * s) for s in (10, 15, 20)] == [501, 751, 1001]


def test_dense_and_onoff_transcripts_admit_exactly_one_ctc_alignment():
    g = torch.Generator().manual_seed(1)
    dense = frame_labels(6, 501, "dense", g)
    assert int(dense.min()) >= 0 and int(dense.max()) < len(TONE_HZ)
    step = (dense[:, 1:] - dense[:, :-1]) % len(TONE_HZ)
    assert int(step.min()) >= 2 and int(step.max()) <= len(TONE_HZ) - 2          # never the same tone, never its neighbour
    assert all(len(transcript(r)) == 501 for r in dense.numpy())                 # T' labels in T' frames: one alignment, no blank
    onoff = frame_labels(4, 501, "onoff", g)
    assert bool((onoff[:, 1::2] == -1).all()) and bool((onoff[:, ::2] == onoff[:, :1]).all())
    # S equal labels need S - 1 blanks between them: 251 + 250 = 501 frames, again one alignment
    assert all(len(t) == 251 and len(set(t)) == 1 for t in (transcript(r) for r in onoff.numpy()))


def test_bursts_keep_two_silent_frames_between_them_and_change_label_every_frame():
    lab = frame_labels(16, 751, "bursts", torch.Generator().manual_seed(2)).numpy()
    for row in lab:
        on = row >= 0
        runs, start = [], None
        for j, v in enumerate(list(on) + [False]):
            if v and start is None:
                start = j
            if not v and start is not None:
                runs.append((start, j)); start = None
        assert runs and all(BURST_FRAMES[0] <= b - a <= BURST_FRAMES[1] for a, b in runs)
        assert all(b2 - a1 >= 2 for (_, a1), (b2, _) in zip(runs[:-1], runs[1:]))    # >= 2 silent frames between bursts
        inside = np.concatenate([(row[a + 1:b] - row[a:b - 1]) % len(TONE_HZ) for a, b in runs])
        assert inside.min() >= 2 and inside.max() <= len(TONE_HZ) - 2
        assert len(transcript(row)) == int(on.sum())                                 # every sounding frame is one label
        assert all(a % SLOT_FRAMES >= 1 and (b - 1) % SLOT_FRAMES <= SLOT_FRAMES - 2 for a, b in runs)


def test_mixed_batches_hold_all_three_kinds_and_clips_are_reproducible():
    lab = frame_labels(32, 501, "mix", torch.Generator().manual_seed(3))
    kinds = ["bursts" if bool((r == -1).any()) and int((r >= 0).sum()) < 400 and len(set(r[r >= 0].tolist())) > 1 else
             ("onoff" if bool((r == -1).any()) else "dense") for r in lab]
    assert kinds[:16] == ["bursts"] * 16 and kinds[16:24] == ["dense"] * 8 and kinds[24:] == ["onoff"] * 8
    w1, l1, t1 = tone_clips(3, 2, 7)
    w2, l2, t2 = tone_clips(3, 2, 7)
    assert torch.equal(w1, w2) and t1 == t2 and l1.tolist() == [32000.0] * 3
    assert tone_clips(3, 2, 8)[2] != t1


def test_a_frame_carries_its_own_tone_and_silence_is_noise_only():
    wav, _, texts = tone_clips(2, 2, 11, noise=0.0)
    lab = frame_labels(2, encoder_frames(32000), "bursts", torch.Generator().manual_seed(11)).numpy()
    assert texts == [transcript(r) for r in lab]
    x = wav[0].numpy()
    for j in range(2, 90):                                                           # frame j = the 320 samples centred on sample 320 j
        seg = x[320 * j - 160: 320 * j + 160]
        if lab[0, j] < 0:
            assert np.abs(seg).max() == 0.0
        else:
            spec = np.abs(np.fft.rfft(seg * np.hanning(320), 4096))
            peak = np.argmax(spec) * 16000.0 / 4096
            assert abs(peak - TONE_HZ[lab[0, j]]) < 40.0, (j, peak, TONE_HZ[lab[0, j]])
