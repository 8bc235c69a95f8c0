"""CPU-only checks of the round-2 host logic: module tree with augmentation, reference-exact mask draws, the fine-tuning
schedule object, the packed-weight cache noticing raw-pointer updates, text encoding (host path) against the reference fixture."""
import numpy as np
import pytest
import torch
from torch import nn


def test_filterbank_tree_with_augmentation():
    from thunder_speech_amd.quartznet.spec_augment import SpecAugment, SpecCutout
    from thunder_speech_amd.quartznet.transform import FilterbankFeatures
    fb = FilterbankFeatures()
    assert len(fb) == 4
    fb = FilterbankFeatures(num_time_masks=2, num_freq_masks=1, mask_time_width=40, mask_freq_width=15)
    assert len(fb) == 5 and isinstance(fb[4].layer[0], SpecAugment)
    sa = fb[4].layer[0]
    assert (sa.time_masks, sa.freq_masks, sa.time_width, sa.freq_width) == (2, 1, 40, 15)
    fb = FilterbankFeatures(num_cutout_masks=3)
    assert isinstance(fb[4].layer[0], SpecCutout) and fb[4].layer[0].rect_masks == 3
    with pytest.raises(ValueError):
        FilterbankFeatures(num_cutout_masks=1, num_time_masks=1)
    assert not list(fb.state_dict()) or set(fb.state_dict()) == {"1.window", "2.layer.0.fb"}


def test_mask_draws_reproduce_the_reference_under_the_same_seed(golden):
    """rng='torch' (default): the table equals what the reference's SpecAugment / SpecCutout drew under the same manual_seed."""
    from thunder_speech_amd.quartznet.spec_augment import SpecAugment, SpecCutout
    g = golden("r2_misc.npz")
    for i in range(3):
        n_time, tw, n_freq, fw = [int(v) for v in g[f"specaug{i}_cfg"]]
        m = SpecAugment(freq_masks=n_freq, time_masks=n_time, freq_width=fw, time_width=tw)
        torch.manual_seed(int(g[f"specaug{i}_seed"]))
        assert np.array_equal(m.draw(64, 301, "cpu").numpy(), g[f"specaug{i}_table"])
    for i in range(2):
        n, tw, fw = [int(v) for v in g[f"cutout{i}_cfg"]]
        m = SpecCutout(rect_masks=n, time_width=tw, freq_width=fw)
        torch.manual_seed(int(g[f"cutout{i}_seed"]))
        assert np.array_equal(m.draw(64, 301, "cpu").numpy(), g[f"cutout{i}_table"])
    assert SpecAugment().draw(64, 301, "cpu") is None
    x = torch.randn(2, 64, 50)
    assert SpecAugment(time_masks=2).eval()(x) is x                  # identity in eval mode, like the reference


def test_text_encode_host_path_matches_reference_fixture(golden):
    from thunder_speech_amd.text_processing.transform import BatchTextTransformer
    g = golden("r2_misc.npz")
    labels = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]
    t1 = BatchTextTransformer(tokens=labels)
    e, l = t1.encode([str(s) for s in g["enc_texts"]])
    assert np.array_equal(e.numpy(), g["enc1"]) and np.array_equal(l.numpy(), g["len1"])
    t2 = BatchTextTransformer(tokens=labels, start_token="<bos>", end_token="<eos>", unknown_token="<unk>")
    e, l = t2.encode([str(s) for s in g["enc_texts_unk"]])
    assert np.array_equal(e.numpy(), g["enc2"]) and np.array_equal(l.numpy(), g["len2"])
    e, l = t1.encode([str(s) for s in g["enc_texts_unk"]])
    assert np.array_equal(e.numpy(), g["enc3"]) and np.array_equal(l.numpy(), g["len3"])
    assert t1.decode_prediction(torch.from_numpy(g["enc1"]), remove_repeated=False) == [str(s) for s in g["dec1"]]


def test_finetune_schedule_freezes_and_unfreezes_like_base_finetuning():
    from thunder_speech_amd.callbacks import FinetuneEncoderDecoder
    from thunder_speech_amd.registry import load_pretrained
    m = load_pretrained("QuartzNet5x5_synthetic")
    cb = FinetuneEncoderDecoder(unfreeze_encoder_at_epoch=2, encoder_initial_lr_div=5, train_batchnorm=True)
    cb.on_fit_start(None, m)
    with pytest.raises(Exception):
        cb.on_fit_start(None, nn.Linear(2, 2))
    m.train()
    cb.freeze_before_training(m)
    bn = [x for x in m.encoder.modules() if isinstance(x, nn.BatchNorm1d)]
    convs = [x for x in m.encoder.modules() if isinstance(x, nn.Conv1d)]
    assert all(p.requires_grad for b in bn for p in b.parameters()) and all(b.training for b in bn)
    assert not any(p.requires_grad for c in convs for p in c.parameters())
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    n0 = len(opt.param_groups)
    cb.finetune_function(m, 1, opt)
    assert len(opt.param_groups) == n0                                  # not yet
    cb.finetune_function(m, 2, opt)
    assert all(p.requires_grad for c in convs for p in c.parameters())
    assert len(opt.param_groups) == n0 + 1 and abs(opt.param_groups[-1]["lr"] - 1e-3 / 5) < 1e-12
    ids = [id(p) for g in opt.param_groups for p in g["params"]]
    assert len(ids) == len(set(ids))                                    # no parameter in two groups
    # train_batchnorm = False: BatchNorm parameters frozen too until the unfreeze.  Like Lightning ^1.7's BaseFinetuning.freeze the
    # schedule flips requires_grad ONLY: every module keeps its train flag (batch statistics, active Dropout)
    m2 = load_pretrained("QuartzNet5x5_synthetic").train()
    cb2 = FinetuneEncoderDecoder(train_batchnorm=False)
    cb2.freeze_before_training(m2)
    assert not any(p.requires_grad for p in m2.encoder.parameters())
    assert all(x.training for x in m2.encoder.modules())
    opt2 = torch.optim.AdamW([p for p in m2.parameters() if p.requires_grad], lr=1e-3)
    cb2.finetune_function(m2, 1, opt2)                                  # train_bn = not train_batchnorm = True: BatchNorm joins too
    assert all(p.requires_grad for p in m2.encoder.parameters())
    assert {id(p) for p in m2.encoder.parameters()} <= {id(p) for g in opt2.param_groups for p in g["params"]}
    assert all(x.training for x in m.encoder.modules())                 # the first schedule left the flags alone as well


def test_packed_cache_sees_version_bumps():
    """optim.FusedAdamW and batch_norm_train update tensors through raw pointers and then bump `_version`; the cache key must
    change with it."""
    from thunder_speech_amd.blocks import _PackedCache
    from thunder_speech_amd.optim import _bump_versions
    w = torch.zeros(4)
    cache, builds = _PackedCache(), []
    build = lambda: builds.append(1) or len(builds)
    assert cache.get([w], build) == 1 and cache.get([w], build) == 1
    _bump_versions([w])
    assert cache.get([w], build) == 2


def test_ctc_rejects_bad_targets_on_the_host():
    from thunder_speech_amd.ctc_loss import _prepare_targets
    t, tl, _ = _prepare_targets(torch.tensor([1, 2, 3, 4, 5]), torch.tensor([2, 3]), 2, 6, "cpu")   # concatenated 1-D form
    assert t.tolist() == [[1, 2, 0], [3, 4, 5]] and tl.tolist() == [2, 3]
    with pytest.raises(ValueError):
        _prepare_targets(torch.tensor([1, 2, 3]), torch.tensor([2, 3]), 2, 6, "cpu")                # lengths do not add up
    with pytest.raises(ValueError):
        _prepare_targets(torch.tensor([[1, 9]]), torch.tensor([2]), 1, 6, "cpu")                     # id outside [0, V)
    t, _, _ = _prepare_targets(torch.tensor([[1, 9]]), torch.tensor([1]), 1, 6, "cpu")               # ... but padding may hold anything
    assert t.tolist() == [[1, 0]]


def test_gradient_sync_hands_a_bucket_view_out_once_and_never_exchanges_a_stale_one():
    """ADVICE (low): a parameter contributing twice in one backward must not get the same bucket view twice (train_ops.grad_out asks
    GradientSync.claim); a parameter without a gradient after a plain optimizer.zero_grad(set_to_none=True) must present zeros, not last
    step's gradient."""
    import torch
    from thunder_speech_amd.parallel import GradientSync
    p, q = torch.nn.Parameter(torch.ones(4)), torch.nn.Parameter(torch.ones(3))
    sync = GradientSync([p, q])
    assert sync.claim(p) and not sync.claim(p) and sync.claim(q)
    sync.zero_grad()
    assert sync.claim(p)                                  # a new step starts with nothing handed out
    ((p * 2).sum() + (q * 3).sum()).backward()
    sync.finish()
    assert torch.equal(p.grad, torch.full((4,), 2.0)) and torch.equal(q.grad, torch.full((3,), 3.0)) and sync.claim(p)
    p.grad, q.grad = None, None                           # what optimizer.zero_grad() (set_to_none=True, the torch default) does
    (p * 5).sum().backward()                              # q takes no part in this step
    sync.finish()
    assert torch.equal(p.grad, torch.full((4,), 5.0)) and torch.equal(q.grad, torch.zeros(3))
    sync.close()


def test_graphed_train_step_refuses_a_capture_without_an_eager_pass():
    import pytest, torch
    from thunder_speech_amd.train_graph import GraphedTrainStep
    with pytest.raises(ValueError, match="warmup"):
        GraphedTrainStep(torch.nn.Linear(1, 1), None, None, warmup=0)


def test_masked_conv_mask_fill_and_length_formula_like_the_reference_test():
    """tests/quartznet/test_blocks_qn.py:135-143 of the reference: mask_fill zeroes the tail, get_seq_len follows the conv formula."""
    import torch
    from thunder_speech_amd.quartznet.blocks import MaskedConv1d
    x = torch.randn(10, 128, 1337)
    lens = torch.Tensor([1000] * 10)
    conv = MaskedConv1d(128, 10, 3)
    x_mask = conv.mask_fill(x, lens)
    assert conv.get_seq_len(lens)[0] == (1000 + 2 * 0 - 1 * (3 - 1) - 1) // 1 + 1
    assert (x_mask[:, :, 1000:] == 0).all() and torch.equal(x_mask[:, :, :1000], x[:, :, :1000])


def test_reference_utils_names(tmp_path):
    """thunder.utils' small helpers (utils.py:32-147) under the same names."""
    import struct, wave
    import pytest
    from thunder_speech_amd.quartznet.compatibility import QuartznetCheckpoint
    from thunder_speech_amd.utils import BaseCheckpoint, audio_len, chain_calls, download_checkpoint, get_files
    assert chain_calls(lambda x: 2 * x, lambda x: 3 * x, lambda x: 4 * x)(1) == 24
    assert QuartznetCheckpoint.from_string("QuartzNet5x5LS_En") is QuartznetCheckpoint.QuartzNet5x5LS_En and issubclass(QuartznetCheckpoint, BaseCheckpoint)
    with pytest.raises(ValueError):
        QuartznetCheckpoint.from_string("nope")
    (tmp_path / "a").mkdir()
    wav = tmp_path / "a" / "x.wav"
    with wave.open(str(wav), "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
        f.writeframes(struct.pack("<8000h", *([0] * 8000)))
    (tmp_path / "b.txt").write_text("t")
    assert get_files(tmp_path, ".wav") == [wav] and abs(audio_len(wav) - 0.5) < 1e-9
    # a bare checkpoint name resolves to the cached `<name>.nemo` -- the file load_quartznet_checkpoint looks for
    (tmp_path / "QuartzNet5x5LS-En.nemo").write_text("w")
    assert download_checkpoint(QuartznetCheckpoint.QuartzNet5x5LS_En, str(tmp_path)) == tmp_path / "QuartzNet5x5LS-En.nemo"
    with pytest.raises(FileNotFoundError):
        download_checkpoint(QuartznetCheckpoint.QuartzNet15x5Base_En, str(tmp_path))


def test_configure_optimizers_follows_the_lightning_contract():
    """module.py:165-189 of the reference: AdamW (or any class) over the parameters that require a gradient; with a scheduler class the
    Lightning dict {"optimizer", "lr_scheduler": {"scheduler", "interval"}}; the magic key "total_steps_arg" names the builder argument that
    receives trainer.estimated_stepping_batches; "interval" is taken out of the scheduler kwargs (default "step")."""
    from types import SimpleNamespace
    from thunder_speech_amd.module import BaseCTCModule
    from thunder_speech_amd.text_processing.transform import BatchTextTransformer
    enc, dec = torch.nn.Linear(4, 4), torch.nn.Linear(4, 3)
    enc.weight.requires_grad_(False)
    text = BatchTextTransformer(tokens=["a", "b"])
    plain = BaseCTCModule(enc, dec, torch.nn.Identity(), text, optimizer_kwargs={"lr": 0.25})
    opt = plain.configure_optimizers()
    assert isinstance(opt, torch.optim.AdamW) and opt.defaults["lr"] == 0.25
    params = [id(p) for g in opt.param_groups for p in g["params"]]
    assert id(enc.weight) not in params and id(enc.bias) in params and id(dec.weight) in params

    sched = BaseCTCModule(enc, dec, torch.nn.Identity(), text, optimizer_class=torch.optim.SGD, optimizer_kwargs={"lr": 0.1},
                          lr_scheduler_class=torch.optim.lr_scheduler.OneCycleLR,
                          lr_scheduler_kwargs={"max_lr": 0.5, "total_steps_arg": "total_steps", "interval": "epoch"})
    fake_trainer = SimpleNamespace(estimated_stepping_batches=321)
    try:
        sched.trainer = fake_trainer                      # plain nn.Module base (no Lightning in this image)
    except Exception:
        object.__setattr__(sched, "_trainer", fake_trainer)
    out = sched.configure_optimizers()
    assert set(out) == {"optimizer", "lr_scheduler"} and isinstance(out["optimizer"], torch.optim.SGD)
    assert out["lr_scheduler"]["interval"] == "epoch"
    assert isinstance(out["lr_scheduler"]["scheduler"], torch.optim.lr_scheduler.OneCycleLR)
    assert out["lr_scheduler"]["scheduler"].total_steps == 321
    assert sched.lr_scheduler_kwargs == {"max_lr": 0.5, "total_steps_arg": "total_steps"}       # the builder kwargs themselves are not consumed


def test_a_subclass_override_of_update_special_optimizer_arg_is_honoured():
    """`_update_special_optimizer_arg` is the reference's hook name (module.py:165-171); configure_optimizers must call the override."""
    from thunder_speech_amd.module import BaseCTCModule
    from thunder_speech_amd.text_processing.transform import BatchTextTransformer

    class Custom(BaseCTCModule):
        def _update_special_optimizer_arg(self, kwargs):
            out = dict(kwargs)
            out["lr"] = out["lr"] * 2
            return out

    m = Custom(torch.nn.Linear(4, 4), torch.nn.Linear(4, 3), torch.nn.Identity(), BatchTextTransformer(tokens=["a", "b"]), optimizer_kwargs={"lr": 0.25})
    assert m.configure_optimizers().defaults["lr"] == 0.5


def test_decode_collapsed_join_plans_equal_the_reference_join():
    """decode_collapsed's three join plans (character table / token table / per-id loop) against the reference semantics restated in
    `_ids_to_text` (text_processing/transform.py:107-120: join, "▁" and "|" -> " ", special-token STRINGS removed), incl. vocabularies whose
    ordinary tokens could spell a special string (they must not take the character plan) and an out-of-vocabulary id (IndexError)."""
    import torch
    from thunder_speech_amd.text_processing.transform import BatchTextTransformer
    labels = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]
    cases = [(labels, {}, "chars"), (labels + ["|", "▁", "é", "ß"], dict(start_token="<s>", end_token="</s>"), "chars"),
             (labels + ["€"], {}, "chars"),
             (labels + ["<"], {}, "table"),                          # "<" + "blank>"-like spellings: the removal pass must see the joined text
             (["▁the", "▁", "a", "b|", "<", "blank", ">", "é", "ab▁c"], {}, "table"),
             (labels, dict(unknown_token="<unk>"), "table"), (["a", "b\x00"], {}, "loop")]
    g = torch.Generator().manual_seed(0)
    for toks, kw, want_plan in cases:
        tt = BatchTextTransformer(tokens=list(toks), **kw)
        assert tt._decode_plan()[0] == want_plan, (toks[-3:], tt._decode_plan()[0])
        v = len(tt.vocab.itos)
        rows = torch.randint(0, v, (9, 203), generator=g, dtype=torch.int32)
        counts = torch.randint(0, 204, (9,), generator=g, dtype=torch.int32)
        counts[0], counts[1] = 0, 203
        for i in range(9):
            rows[i, int(counts[i]):] = 0                              # what ts_greedy_decode leaves beyond the count
        assert tt.decode_collapsed(rows, counts) == [tt._ids_to_text(rows[i, : int(counts[i])].tolist()) for i in range(9)]
        rows[3, 0] = v
        counts[3] = max(int(counts[3]), 1)
        with pytest.raises(IndexError):
            tt.decode_collapsed(rows, counts)
