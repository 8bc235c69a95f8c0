"""Round-6 GPU tests: BASELINE.json configs[0] at its named size (QuartzNet5x5 predict() on 4 x 10 s) against the fixture made from the real
reference modules; `predict()` / `forward()` replayed from the module's per-signature hipGraph (same results as eager launches, invalidated
by weight updates, outputs never aliased); bench.py's multi-rank code path run with ONE rank under torch.distributed.run (RCCL group,
probe all-reduce, barriers, c4_ddp's exchange, destroy); a block output with a second consumer fails loudly in the training path."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import decode as odec, frontend as ofe, tcs as otcs

pytestmark = pytest.mark.gpu
LABELS = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]


def _c1_module_and_input(golden):
    from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
    g = golden("qn5x5_c1_4x10s.npz")
    arch = otcs.quartznet_arch(repeat_blocks=1)
    sd = otcs.synth_encoder_state(arch, seed=int(g["enc_seed"]), calibrate=True)
    dsd = {"weight": torch.from_numpy(g["dec_weight"]), "bias": torch.from_numpy(g["dec_bias"])}
    module = build_synthetic_quartznet(repeat_blocks=1, encoder_state=sd, decoder_state=dsd).cuda().eval()
    rng = np.random.Generator(np.random.PCG64(int(g["wav_seed"])))
    wav = torch.from_numpy((0.1 * rng.standard_normal(tuple(g["wav_shape"]))).astype(np.float32))
    return g, arch, sd, dsd, module, wav


def test_c1_quartznet5x5_predict_4x10s_strings_equal_the_reference(golden):
    """configs[0] at its NAMED size, through the plugin surface: `module.predict(wav)` on 4 x 10 s clips returns exactly the strings the
    reference's FilterbankFeatures -> QuartznetEncoder -> conv1d_decoder -> argmax -> decode_prediction produced (tests/golden/
    make_golden_c1.py; decoder fitted to a top-1 / top-2 margin of 4 on EVERY frame).  Also: every frame's argmax equals the reference's,
    the logits agree with the oracle's (itself within 2e-3 of the reference when the fixture was made) to well inside half the margin, and
    the fixture's sampled reference logit columns agree with the oracle on this box."""
    g, arch, sd, dsd, module, wav = _c1_module_and_input(golden)
    want = [str(s) for s in g["strings"]]
    x = wav.cuda()
    with torch.no_grad():
        eager = module.predict(x)                                          # first sighting of the signature: eager launches
        assert module._infer_graphs[1].count() == 0
        graphed = module.predict(x)                                        # second: captured and replayed
        assert module._infer_graphs[1].count() == 1
        again = module.predict(x)
        logits, out_len = module(x, torch.full((4,), float(x.shape[1]), device="cuda"))
    assert eager == want and graphed == want and again == want
    assert np.array_equal(out_len.cpu().numpy(), g["out_lengths"])
    got = logits.float().cpu().numpy()
    assert np.array_equal(got.argmax(1), g["labels"].astype(np.int64))     # every frame, no "decided" mask
    lengths = torch.full((4,), float(wav.shape[1]))
    feats, fl = ofe.filterbank_features(wav, lengths)
    enc, _ = otcs.encoder_forward(arch, sd, feats, fl)
    ref = otcs.conv1d_decoder_forward(dsd, enc).numpy()
    cols = g["sample_cols"]
    assert np.abs(ref[:, :, cols] - g["logits_sample"]).max() <= 2e-3      # the oracle on this box vs the reference's stored columns
    margin = float(g["min_margin"])
    assert np.abs(got - ref).max() <= 0.45 * margin                        # stated tolerance of the bf16 path here: < half the smallest margin
    assert odec.decode_prediction(odec.argmax_classes(ref), odec.Vocab(list(LABELS))) == want


def test_inference_graph_matches_eager_is_invalidated_by_weight_updates_and_never_aliases_outputs(golden):
    g, arch, sd, dsd, module, wav = _c1_module_and_input(golden)
    x = wav[:2, :48000].contiguous().cuda()
    lens = torch.tensor([48000.0, 31000.0], device="cuda")
    with torch.no_grad():
        module.graph_inference = False
        ref_logits, ref_len = module(x, lens)
        module.graph_inference = None
        module.reset_inference_graphs()
        a0, _ = module(x, lens)                                            # eager (first sighting)
        a1, l1 = module(x, lens)                                           # captured + replayed
        assert module._infer_graphs[1].count() == 1
        assert torch.equal(a0, ref_logits) and torch.equal(a1, ref_logits) and torch.equal(l1, ref_len)
        x2 = (0.5 * x).contiguous()
        b1, _ = module(x2, lens)                                           # same signature at another address: not this graph's input
        assert torch.equal(a1, ref_logits), "a returned tensor was overwritten by the next call"
        module.graph_inference = False
        b_ref, _ = module(x2, lens)
        module.graph_inference = None
        assert torch.equal(b1, b_ref)
        # an in-place weight update (what an optimizer step or load_state_dict does) must drop the graph
        module.decoder.bias.add_(3.0 * torch.arange(29, device="cuda", dtype=torch.float32))
        c0, _ = module(x, lens)
        c1, _ = module(x, lens)
        module.graph_inference = False
        c_ref, _ = module(x, lens)
        assert torch.equal(c0, c_ref) and torch.equal(c1, c_ref) and not torch.equal(c_ref, ref_logits)
        # the same shape arriving at NEW addresses every time: from the fourth sighting on, one copying graph serves them all
        module.graph_inference = None
        module.reset_inference_graphs()
        keep, outs = [], []
        for k in range(6):
            xk = (x * (1.0 + 0.1 * k)).contiguous()
            keep.append(xk)                                                # held, so that every call sees a fresh address
            outs.append(module(xk, lens)[0])
        assert module._infer_graphs[1].count() == 0 and module._infer_graphs[2].count() == 1
        module.graph_inference = False
        for xk, got in zip(keep, outs):
            assert torch.equal(got, module(xk, lens)[0])
    # under autograd / in train mode the graph path is never taken
    module.graph_inference = None
    module.reset_inference_graphs()
    module(x, lens); module(x, lens)
    assert module.__dict__.get("_infer_graphs") is None
    # a caller that captures its own graph launches into it (no nested replay)
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            module(x, lens); module(x, lens)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                out, _ = module(x, lens)
        torch.cuda.current_stream().wait_stream(side)
        gr.replay()
        torch.cuda.synchronize()
        module.graph_inference = False
        want, _ = module(x, lens)
        assert torch.equal(out, want)


def test_bench_multi_rank_path_runs_with_one_rank_under_torch_distributed_run(tmp_path):
    """The `WORLD_SIZE`-set branch of bench.py on the hardware at hand: launched by torch.distributed.run with ONE rank it brings up the RCCL
    group (device_id), runs the probe all-reduce, the barriers, max-over-ranks, c4_ddp with GradientSync's reduce-scatter / all-gather over the
    real process group, and destroy_process_group.  One JSON line, rc 0, rccl_world_size from the all-reduce, c4_ddp without error."""
    from thunder_speech_amd.parallel import launch_ranks
    out_path, err_path = tmp_path / "bench.out", tmp_path / "bench.err"
    env = dict(os.environ, TS_BENCH_EXTRA_DEADLINE_S="600")
    with open(out_path, "w") as fo, open(err_path, "w") as fe:
        rc = launch_ranks(os.path.join(ROOT, "bench.py"), 1, ["--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                                                               "--no-predict-api", "--extra", "c4_ddp"], timeout_s=900, env=env, stdout=fo, stderr=fe)
    err = err_path.read_text()
    assert rc == 0, err[-3000:]
    lines = [ln for ln in out_path.read_text().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res["n_gpus"] == 1 and res["rccl_world_size"] == 1 and res["process_group"] == "nccl (RCCL), size from a real all-reduce"
    ddp = res["extra"]["c4_ddp"]
    assert "error" not in ddp, ddp
    assert ddp["process_group"] == "torch.distributed.run" and ddp["n_collectives_per_step"] >= 2
    assert ddp["ms_per_step"] > 0 and np.isfinite(ddp["loss_first_last"]).all()
    assert res["scaling_block"]["n_gpus"] == 1 and res["scaling_block"]["c4_ddp_step_per_s"] == ddp["value"]


def test_block_output_with_a_second_consumer_fails_loudly_instead_of_dropping_the_residual_gradient():
    """Fork.backward parks (g1, g2) for the previous block's one-launch tail.  If that block output has another consumer
    (hook / auxiliary loss) autograd sums g1 into a new buffer, the key misses and the residual-branch gradient used to vanish silently.  Now
    the end-of-backward check raises; with DEFER_FORK_ADD off the same graph gives the sum of the two losses' gradients."""
    from thunder_speech_amd import train_ops as T
    from thunder_speech_amd.quartznet.blocks import QuartznetBlock
    torch.manual_seed(0)
    b1 = QuartznetBlock(16, 16, repeat=2, kernel_size=(5,), separable=True, dropout=0.0).cuda().train()
    b2 = QuartznetBlock(16, 16, repeat=2, kernel_size=(5,), separable=True, dropout=0.0).cuda().train()
    x0 = torch.randn(4, 16, 100, device="cuda")
    lens = torch.tensor([100, 90, 80, 70], device="cuda")
    params = list(b1.parameters()) + list(b2.parameters())

    def run(aux: bool, main: bool = True):
        for p in params:
            p.grad = None
        x = x0.clone().requires_grad_(True)
        y1, l1 = b1(x, lens)
        y2, _ = b2(y1, l1)
        assert getattr(y1, "_ts_tail_out", False), "the test must exercise the deferred add"
        loss = 0.0
        if main:
            loss = loss + (T.from_act(y2) ** 2).sum()
        if aux:
            loss = loss + (T.from_act(y1) * 0.37).sum()
        loss.backward()
        return x.grad.clone(), [None if p.grad is None else p.grad.clone() for p in params]

    g_main, p_main = run(False)
    assert not T._PENDING_ADD
    with pytest.raises(RuntimeError, match="parked gradient"):
        run(True)
    assert not T._PENDING_ADD
    g_again, _ = run(False)                                                  # the failure leaves nothing behind
    assert torch.equal(g_again, g_main)
    T.DEFER_FORK_ADD = False
    try:
        g_both, p_both = run(True)
        g_aux, p_aux = run(True, main=False)
    finally:
        T.DEFER_FORK_ADD = True
    want = g_main + g_aux
    assert float((g_both - want).abs().max()) <= 2e-4 * float(want.abs().max())
    for pb, pm, pa in zip(p_both, p_main, p_aux):
        if pb is None:
            continue
        w = (pm if pm is not None else 0) + (pa if pa is not None else 0)
        assert float((pb - w).abs().max()) <= 5e-4 * max(float(torch.as_tensor(w).abs().max()), 1e-3)


@pytest.mark.parametrize("train", [False, True])
def test_front_end_register_prefetch_equals_the_direct_staging_bit_for_bit(train):
    """stft_mel_kernel prefetches the next frame group's samples with `asm volatile` loads whose wait the compiler does not track
    (csrc/frontend.hip fetch / commit); a toolchain that re-coloured those registers in between would produce silently wrong features.  The same
    samples at a 4-byte-misaligned address make every group take the plain `stage_direct` path instead: the two feature tensors must be
    IDENTICAL, in eval mode and in train mode (dither drawn from the same seed)."""
    from thunder_speech_amd import rng
    from thunder_speech_amd.quartznet.transform import FilterbankFeatures
    fb = FilterbankFeatures().cuda()
    fb.train(train)
    n = 16000 * 7
    g = torch.Generator().manual_seed(9)
    base = torch.zeros(3 * n + 8, device="cuda")
    wav = (0.1 * torch.randn(3, n, generator=g)).cuda()
    aligned = wav.clone()
    shifted = base[1: 1 + 3 * n].view(3, n)
    shifted.copy_(wav)
    assert aligned.data_ptr() % 16 == 0 and shifted.data_ptr() % 16 == 4
    lengths = torch.tensor([n, n - 1234, n // 2], device="cuda", dtype=torch.int32)
    outs = []
    for x in (aligned, shifted):
        if train:
            torch.manual_seed(77)                      # rng.next_seed draws the dither seed from torch's CPU generator
        feats, flen = fb(x, lengths)
        outs.append((feats.clone(), flen.clone(), fb.last_logmel().clone()))
    assert torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][2], outs[1][2]), "log-mel differs between the prefetching and the direct staging path"
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1].dtype == torch.int64 and outs[0][1].tolist() == [n // 160 + 1, (n - 1234) // 160 + 1, (n // 2) // 160 + 1]


@pytest.mark.parametrize("b,ci,co,t", [(4, 512, 1024, 751), (3, 512, 640, 333)])
def test_wide_frame_pointwise_tile_equals_the_default_tile_bit_for_bit(b, ci, co, t):
    """ts_tcs_pointwise_wide(1): consumer waves of 192 frames x 32 channels for the tail-zero pointwise-only launches (measured: no faster,
    profiles/round6_tcs_256.txt, so the default stays 0) -- same accumulation order, so the same bits."""
    from thunder_speech_amd import _lib, plan, tensors as TS
    L = _lib.lib()
    g = torch.Generator().manual_seed(b * 1000 + co)
    w = torch.randn(co, ci, 1, generator=g) / ci ** 0.5
    bn = [torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1, torch.rand(co, generator=g) + 0.5]
    layer = plan.make_tcs_layer("cuda", dw_w=None, pw_w=w, bn=bn, kernel=1, stride=1, dilation=1, padding=0, relu=True)
    lens = torch.randint(t // 2, t + 1, (b,), generator=g).to(torch.int32).cuda()
    lens[0] = t
    xb = TS.arena(("wide_x", 0), b, ci, t, "cuda")
    xb.zero_()
    xb[:, :, :t] = torch.randn(b, ci, t, generator=g).to(torch.bfloat16).cuda()
    for i in range(b):
        xb[i, :, int(lens[i]):] = 0
    outs = []
    try:
        for wide in (0, 1):
            assert L.ts_tcs_pointwise_wide(wide) in (0, 1)
            out = TS.arena(("wide_o", wide), b, co, t, "cuda")
            out.fill_(7.0)
            layer.run(xb, t, lens, out=out, in_tail_zero=True, zero_tail=True)
            torch.cuda.synchronize()
            outs.append(out[:, :, :t].clone())
    finally:
        assert L.ts_tcs_pointwise_wide(0) == 1
    assert torch.equal(outs[0], outs[1])
    assert float(outs[0].float().abs().sum()) > 0
    for i in range(b):
        assert float(outs[1][i, :, int(lens[i]):].float().abs().sum()) == 0
