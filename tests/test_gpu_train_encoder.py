"""GPU parity of the training-mode encoder ops (fine-tuning with the encoder unfrozen): forward outputs and running-stat
updates vs fixtures produced by the REAL reference blocks in train mode, gradients vs torch autograd through the oracle."""
import numpy as np
import pytest
import torch

from conftest import sd_from_npz
from oracle import tcs as otcs

pytestmark = pytest.mark.gpu

CASES = {
    "qn_res_k11": dict(in_ch=16, out_ch=32, repeat=3, kernel=11),
    "qn_stride2": dict(in_ch=16, out_ch=24, repeat=1, kernel=33, stride=2, residual=False),
    "qn_dil2": dict(in_ch=24, out_ch=24, repeat=1, kernel=13, dilation=2, residual=False),
    "qn_dense_k1": dict(in_ch=24, out_ch=48, repeat=1, kernel=1, residual=False, separable=False),
}


def _block(spec, sd):
    from thunder_speech_amd.quartznet.blocks import QuartznetBlock
    blk = QuartznetBlock(spec.in_ch, spec.out_ch, repeat=spec.repeat, kernel_size=(spec.kernel,), stride=(spec.stride,),
                         dilation=(spec.dilation,), residual=spec.residual, separable=spec.separable)
    blk.load_state_dict(sd, strict=True)
    return blk.cuda().train()


@pytest.mark.parametrize("name", sorted(CASES))
def test_train_forward_and_running_stats_match_reference_fixture(golden, name):
    g = golden("blocks.npz")
    spec = otcs.BlockSpec(**CASES[name])
    sd = sd_from_npz(g, f"{name}/sd/")
    x, lengths = torch.from_numpy(g[f"{name}/x"]), torch.from_numpy(g[f"{name}/lengths"])
    blk = _block(spec, sd)
    y, yl = blk(x.cuda(), lengths.cuda())
    assert np.array_equal(yl.cpu().numpy(), g[f"{name}/out_lengths"])
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"{name}/y_train"], atol=2e-4)
    new = blk.state_dict()
    for k in sd:
        if "running_" in k:
            np.testing.assert_allclose(new[k].cpu().numpy(), g[f"{name}/new/" + k.replace(".", "/")], atol=2e-5)
        if k.endswith("num_batches_tracked"):
            assert int(new[k]) == int(sd[k]) + 1


@pytest.mark.parametrize("name", sorted(CASES))
def test_train_backward_matches_autograd_through_the_oracle(golden, name):
    g = golden("blocks.npz")
    spec = otcs.BlockSpec(**CASES[name])
    sd = sd_from_npz(g, f"{name}/sd/")
    x, lengths = torch.from_numpy(g[f"{name}/x"]), torch.from_numpy(g[f"{name}/lengths"])
    gen = torch.Generator().manual_seed(1)
    # CPU: autograd through the oracle's train-mode block
    sd_ref = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    x_ref = x.clone().requires_grad_(True)
    y_ref, _ = otcs.block_forward(spec, sd_ref, "", x_ref, lengths, training=True)
    cot = torch.randn(y_ref.shape, generator=gen)
    (y_ref * cot).sum().backward()
    # GPU
    blk = _block(spec, sd)
    xg = x.clone().cuda().requires_grad_(True)
    y, _ = blk(xg, lengths.cuda())
    (y * cot.cuda()).sum().backward()
    sx = float(x_ref.grad.abs().max())
    assert float((xg.grad.cpu() - x_ref.grad).abs().max()) <= 2e-3 * max(sx, 1e-3)
    for k, p in blk.named_parameters():
        want = sd_ref[k].grad
        assert want is not None, k
        s = max(float(want.abs().max()), 1e-3)
        assert float((p.grad.cpu() - want).abs().max()) <= 2e-3 * s, k


def test_unfrozen_finetune_step_decreases_loss():
    """QuartzNet5x5, everything trainable: training_step -> backward through decoder, 6 blocks, BN -> FusedAdamW."""
    from thunder_speech_amd.optim import FusedAdamW
    from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
    arch = otcs.quartznet_arch(repeat_blocks=1)
    m = build_synthetic_quartznet(repeat_blocks=1, encoder_state=otcs.synth_encoder_state(arch, seed=0, calibrate=True),
                                  decoder_state=otcs.synth_decoder_state(1024, 29, seed=1)).cuda().train()
    g = torch.Generator().manual_seed(9)
    wav = (0.1 * torch.randn(4, 24000, generator=g)).cuda()
    lengths = torch.tensor([24000.0, 20000.0, 16000.0, 24000.0]).cuda()
    texts = ["abc", "hello", "data", "test"]
    opt = FusedAdamW(m.parameters(), lr=2e-3, weight_decay=0.0)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = m.training_step((wav, lengths, texts), 0)
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses


def test_frozen_schedule_runs_the_encoder_in_train_mode_and_trains_batchnorm_only():
    """What the reference's first fine-tuning phase really is (callbacks.py:56-62 -> Lightning ^1.7 BaseFinetuning.freeze): the
    encoder's conv weights stop receiving gradients, its BatchNorm parameters keep them (train_batchnorm=True), and NO module leaves
    train mode -- so the step is a train-mode forward (batch statistics, running-stat updates) and a backward through the whole
    encoder that reaches the BatchNorm parameters."""
    from torch import nn
    from thunder_speech_amd.callbacks import FinetuneEncoderDecoder
    from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
    arch = otcs.quartznet_arch(repeat_blocks=1)
    m = build_synthetic_quartznet(repeat_blocks=1, encoder_state=otcs.synth_encoder_state(arch, seed=0, calibrate=True),
                                  decoder_state=otcs.synth_decoder_state(1024, 29, seed=1)).cuda().train()
    FinetuneEncoderDecoder(train_batchnorm=True).freeze_before_training(m)
    bns = [x for x in m.encoder.modules() if isinstance(x, nn.BatchNorm1d)]
    convs = [x for x in m.encoder.modules() if isinstance(x, nn.Conv1d)]
    rm0 = [b.running_mean.clone() for b in bns]
    g = torch.Generator().manual_seed(9)
    wav = (0.1 * torch.randn(3, 24000, generator=g)).cuda()
    lengths = torch.tensor([24000.0, 20000.0, 16000.0]).cuda()
    loss = m.training_step((wav, lengths, ["abc", "hello", "data"]), 0)
    loss.backward()
    assert torch.isfinite(loss)
    assert all(c.weight.grad is None for c in convs)
    assert all(b.weight.grad is not None and torch.isfinite(b.weight.grad).all() and float(b.weight.grad.abs().max()) > 0 for b in bns)
    assert m.decoder.weight.grad is not None
    assert all(not torch.equal(a, b.running_mean) for a, b in zip(rm0, bns))             # train-mode BatchNorm: statistics moved
    assert all(int(b.num_batches_tracked) == 1 for b in bns)


@pytest.mark.parametrize("act", ["fp32", "bf16"])
def test_frozen_convolutions_leave_every_other_gradient_unchanged(act):
    """With the convolution weights frozen the backward pass takes the data-gradient-only forms (no 1x1 weight-gradient GEMM, the depthwise
    backward with a null weight-gradient pointer): the gradients that ARE still computed -- BatchNorm parameters, decoder -- and the loss
    must be exactly those of the fully trainable step (same kernels for the data path, same order of operations)."""
    from torch import nn
    from thunder_speech_amd import train_ops
    from thunder_speech_amd.callbacks import FinetuneEncoderDecoder
    from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
    arch = otcs.quartznet_arch(repeat_blocks=1)
    g = torch.Generator().manual_seed(3)
    wav = (0.1 * torch.randn(4, 32000, generator=g)).cuda()
    lengths = torch.tensor([32000.0, 30000.0, 20000.0, 16000.0]).cuda()
    texts = ["abc", "hello world", "data", "xy"]
    grads = {}
    train_ops.set_activation_dtype(act)
    try:
        for frozen in (False, True):
            m = build_synthetic_quartznet(repeat_blocks=1, encoder_state=otcs.synth_encoder_state(arch, seed=0, calibrate=True),
                                          decoder_state=otcs.synth_decoder_state(1024, 29, seed=1)).cuda().train()
            if frozen:
                FinetuneEncoderDecoder(train_batchnorm=True).freeze_before_training(m)
            torch.manual_seed(1234)                      # the dither / SpecAugment seeds of the step come from torch's CPU generator (rng.py)
            loss = m.training_step((wav, lengths, texts), 0)
            loss.backward()
            grads[frozen] = (float(loss), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    finally:
        train_ops.set_activation_dtype("fp32")
    (l0, g0), (l1, g1) = grads[False], grads[True]
    assert l0 == l1
    assert g1 and set(g1) < set(g0) and not any(k.endswith("conv.weight") for k in g1)
    for k, v in g1.items():
        assert torch.equal(v, g0[k]), k


def test_batchnorm_in_eval_mode_inside_a_training_block_is_refused():
    """the training kernels always normalise with batch statistics; a BatchNorm1d that was switched to eval() while
    its block trains must not silently do that (nn.BatchNorm1d would use -- and keep -- its running statistics)."""
    from torch import nn
    from thunder_speech_amd.quartznet.blocks import QuartznetBlock
    blk = QuartznetBlock(64, 64, repeat=2, kernel_size=(11,), separable=True).cuda().train()
    bn = next(x for x in blk.modules() if isinstance(x, nn.BatchNorm1d))
    bn.eval()
    rm = bn.running_mean.clone()
    x = torch.randn(2, 64, 100, device="cuda")
    with pytest.raises(NotImplementedError):
        blk(x, torch.tensor([100, 60], device="cuda"))
    assert torch.equal(rm, bn.running_mean)


@pytest.mark.parametrize("name", sorted(CASES))
def test_bf16_activation_mode_stays_within_bf16_tolerance(golden, name):
    """train_ops.set_activation_dtype("bf16") (mixed precision): activations and their gradients stored as bf16, f32 arithmetic in
    every kernel, bf16 GEMM operands with f32 accumulation; parameters, their gradients and the BatchNorm statistics f32.
    Stated tolerance vs the REAL reference's train-mode output (fixture): 3 % of the output scale (a bf16 ulp is 0.4 %, a
    repeat stores three rounded tensors); gradients vs the f32 path of this library: rms within 5 % per tensor, largest entry
    within 35 % (tiny fixtures: BatchNorm statistics over a few hundred frames, ReLU gates that flip)."""
    from thunder_speech_amd import train_ops
    g = golden("blocks.npz")
    spec = otcs.BlockSpec(**CASES[name])
    sd = sd_from_npz(g, f"{name}/sd/")
    x, lengths = torch.from_numpy(g[f"{name}/x"]), torch.from_numpy(g[f"{name}/lengths"])
    cot = torch.randn(g[f"{name}/y_train"].shape, generator=torch.Generator().manual_seed(1)).cuda()

    def run():
        blk = _block(spec, sd)
        xg = x.clone().cuda().requires_grad_(True)
        y, _ = blk(xg, lengths.cuda())
        (y.float() * cot).sum().backward()
        return y.detach().float(), xg.grad, {k: p.grad for k, p in blk.named_parameters()}, blk

    y0, gx0, gp0, _ = run()
    train_ops.set_activation_dtype("bf16")
    try:
        y1, gx1, gp1, blk = run()
        assert blk(x.cuda(), lengths.cuda())[0].dtype == torch.bfloat16
    finally:
        train_ops.set_activation_dtype("fp32")
    ref = torch.from_numpy(g[f"{name}/y_train"]).cuda()
    scale = float(ref.abs().max())
    assert float((y1 - ref).abs().max()) <= 0.03 * scale and float((y0 - ref).abs().max()) <= 2e-4
    rel = lambda a, b: float((a - b).abs().max()) / max(float(b.abs().max()), 1e-3)
    rms = lambda a, b: float((a - b).pow(2).mean().sqrt()) / max(float(b.pow(2).mean().sqrt()), 1e-6)
    assert gx1.dtype == torch.float32 and rel(gx1, gx0) <= 0.35 and rms(gx1, gx0) <= 0.15
    for k in gp0:
        assert gp1[k].dtype == torch.float32 and rel(gp1[k], gp0[k]) <= 0.35 and rms(gp1[k], gp0[k]) <= 0.2, (k, rel(gp1[k], gp0[k]), rms(gp1[k], gp0[k]))
    assert not torch.equal(y1, y0)                      # the mode really changes the arithmetic


@pytest.mark.parametrize("act", ["fp32", "bf16"])
@pytest.mark.parametrize("batch,ch,t,k,dil", [(1, 2, 37, 5, 1), (5, 6, 501, 33, 1), (3, 4, 512, 75, 1), (2, 2, 1100, 127, 1), (33, 2, 64, 11, 1),
                                               (2, 8, 1536, 39, 1), (3, 4, 501, 87, 2), (2, 3, 1100, 33, 2), (33, 2, 64, 11, 2), (1, 1, 37, 5, 2),
                                               (2, 2, 2050, 127, 2), (32, 4, 501, 63, 1), (20, 2, 1100, 127, 1), (17, 6, 90, 5, 1), (32, 2, 501, 75, 1),
                                               (40, 2, 333, 39, 1)])
def test_same_depthwise_pair_kernels_match_conv1d_autograd(act, batch, ch, t, k, dil):
    """The packed-FMA "same" depthwise kernels (one wavefront = two channel rows, csrc/train_enc.hip dw_fwd_pair / dw_bwd_pair)
    against F.conv1d(groups = C) + autograd on the masked input (quartznet/blocks.py:169-182): ragged lengths, several 512-frame
    wave tiles, clip counts that leave waves idle; dilation 2 (the K87 block of QuartzNet) runs the same kernels in phase-split form
    (a row = (even, odd) frame pairs; odd channel counts allowed); with bf16 rows and >= 17 clips the forward runs on the matrix cores
    (dw_fwd_mfma_kernel: Toeplitz x clips, taps rounded to bf16).  fp32 rows: 2e-5 of the scale; bf16 rows: inputs are rounded once (so the
    reference sees the same values) and the outputs once more: 1 % of the scale."""
    from thunder_speech_amd import train_ops as T
    g = torch.Generator().manual_seed(batch * 1000 + t + k + dil)
    x = torch.randn(batch, ch, t, generator=g)
    w = torch.randn(ch, 1, k, generator=g) / k ** 0.5
    cot = torch.randn(batch, ch, t, generator=g)
    lengths = torch.randint(max(1, t // 3), t + 1, (batch,), generator=g)
    lengths[0] = t
    if act == "bf16":
        x, cot = x.bfloat16().float(), cot.bfloat16().float()
    mask = (torch.arange(t)[None, :] < lengths[:, None])[:, None, :]
    xr, wr = (x * mask).double().requires_grad_(True), w.double().requires_grad_(True)
    yr = torch.nn.functional.conv1d(xr, wr, padding=dil * (k - 1) // 2, dilation=dil, groups=ch) * mask        # len_out = len_in: re-masked output
    (yr * cot.double()).sum().backward()

    T.set_activation_dtype(act)
    try:
        xg = x.cuda().requires_grad_(True)
        wg = w.cuda().requires_grad_(True)
        li = lengths.to(torch.int32).cuda()
        y = T.DepthwiseConv.apply(T.to_act(xg), wg, li, k, 1, dil, dil * (k - 1) // 2, li)
        assert y.dtype == (torch.bfloat16 if act == "bf16" else torch.float32)
        (T.from_act(y) * cot.cuda()).sum().backward()
    finally:
        T.set_activation_dtype("fp32")
    tol = 1e-2 if act == "bf16" else 2e-5
    for name, got, ref in (("y", T.from_act(y).detach(), yr), ("dx", xg.grad * mask.cuda(), xr.grad * mask), ("dw", wg.grad, wr.grad)):
        ref = ref.detach().float().cuda()
        err = float((got.float() - ref).abs().max())
        assert err <= tol * max(float(ref.abs().max()), 1.0), (name, err, float(ref.abs().max()))


@pytest.mark.parametrize("act,optimizer,segments", [("fp32", "fused", 1), ("bf16", "fused", 1), ("bf16", "torch", 1), ("fp32", "fused", 3), ("bf16", "fused", 3)])
def test_graphed_training_step_follows_the_eager_step(act, optimizer, segments):
    """train_graph.GraphedTrainStep (features -> encoder -> decoder -> CTC -> backward replayed from ONE hipGraph, front end /
    exchange / optimizer outside) against the same steps launched eagerly (module.py:102-113 + backward + GradientSync.finish +
    FusedAdamW): same seeds, so the dither draws are the same too; the first loss agrees to 1e-6 relative, the later ones to 1 % (fp32; the
    weight gradients of the depthwise convolutions leave through float atomics, whose order is not fixed, and the loss falls 5x in
    these 4 steps) / 5 % (bf16 activations), the 4-step parameter update to 20 % / 50 % in L2 (AdamW turns noise-level gradient entries into full +-lr steps), and the BatchNorm running statistics count exactly 4 steps: the warm-up and
    capture passes leave no trace.
    segments=3: the same step replayed as THREE graphs (backward pass cut at two block boundaries, train_graph.SegmentedBackward) with
    GradientSync's buckets = the pieces and each bucket sent through its loop-back exchange (pack -> [one rank: no bytes move] -> unpack on
    the side stream) while the next piece replays: same numbers (the bf16 wire rounds the gradients once, inside the bf16 tolerance)."""
    from thunder_speech_amd import train_ops
    from thunder_speech_amd.optim import FusedAdamW
    from thunder_speech_amd.parallel import GradientSync
    from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
    from thunder_speech_amd.train_graph import GraphedTrainStep, segment_parameters
    arch = otcs.quartznet_arch(repeat_blocks=1)
    g = torch.Generator().manual_seed(9)
    wavs = [(0.1 * torch.randn(4, 24000, generator=g)).cuda() for _ in range(2)]
    lengths = torch.tensor([24000.0, 20000.0, 16000.0, 24000.0]).cuda()
    texts = [["abc", "hello", "data", "test"], ["speech", "to", "text x", "qrs"]]

    def make(segmented=False):
        m = build_synthetic_quartznet(repeat_blocks=1, encoder_state=otcs.synth_encoder_state(arch, seed=0, calibrate=True),
                                      decoder_state=otcs.synth_decoder_state(1024, 29, seed=1)).cuda().train()
        params = [p for p in m.parameters() if p.requires_grad]
        # "torch": a plain torch optimizer -- the graph reads derived weight copies (MFMA fragments) that only FusedAdamW refreshes by
        # itself; GraphedTrainStep has to bring them up to date after ANY optimizer step
        opt = FusedAdamW(params, lr=1e-3, weight_decay=0.0) if optimizer == "fused" else torch.optim.AdamW(params, lr=1e-3, weight_decay=0.0)
        if segmented:
            return m, opt, GradientSync(params, groups=segment_parameters(m, segments), loopback=True)
        return m, opt, GradientSync(params, loopback=segments > 1)      # the eager run rounds its gradients through the same wire

    train_ops.set_activation_dtype(act)
    try:
        torch.manual_seed(5)
        m0, opt0, sync0 = make()
        eager = []
        for i in range(4):
            sync0.zero_grad()
            loss = m0.training_step((wavs[i % 2], lengths, texts[i % 2]), 0)
            loss.backward()
            sync0.finish()
            opt0.step()
            eager.append(float(loss.detach()))
        sync0.close()
        torch.manual_seed(5)
        m1, opt1, sync1 = make(segmented=segments > 1)
        step = GraphedTrainStep(m1, opt1, sync1, max_target_len=16, segments=segments)
        n_buckets = len(sync1.buckets)
        graphed = [float(step((wavs[i % 2], lengths, texts[i % 2]))) for i in range(4)]
        if optimizer == "fused":
            # a segmented step updates bucket k's parameters on the side stream right behind bucket k's exchange: every parameter still takes exactly
            # one optimizer step per training step
            assert all(opt1.state[p]["step"] == 4 for grp in opt1.param_groups for p in grp["params"])
        sync1.close()
    finally:
        train_ops.set_activation_dtype("fp32")
    assert step.replays == 4 and len(step._graphs) == 1
    if segments > 1:
        assert step.segments == segments == n_buckets and len(next(iter(step._graphs.values()))[0]) == segments
        assert sync1.n_collectives == 4 * segments * (2 if sync1.collective == "reduce_scatter" else 1)      # every bucket went through its exchange
    # the first step sees identical parameters; afterwards the two runs drift apart through the atomics' summation order, amplified
    # by a loss that falls 5x in 4 steps
    assert abs(eager[0] - graphed[0]) <= (1e-3 if act == "bf16" else 1e-6) * abs(eager[0]), (eager, graphed)
    tol = 5e-2 if act == "bf16" else 1e-2
    for a, b in zip(eager, graphed):
        assert abs(a - b) <= tol * abs(a), (eager, graphed)
    assert eager[-1] != eager[0]
    # AdamW turns noise-level gradient entries into +-lr steps, so single entries may differ by a step; the UPDATE as a whole agrees
    ref, _, sref = make()
    sref.close()
    du0 = torch.cat([(p0 - r).flatten() for p0, r in zip(m0.parameters(), ref.parameters())])
    du1 = torch.cat([(p1 - r).flatten() for p1, r in zip(m1.parameters(), ref.parameters())])
    assert float((du0 - du1).norm() / du0.norm()) <= (0.5 if act == "bf16" else 0.2)
    for (k, b0), b1 in zip(m0.named_buffers(), m1.buffers()):
        if k.endswith("num_batches_tracked"):
            assert int(b0) == int(b1) == 4, k
        elif k.endswith("running_mean") or k.endswith("running_var"):
            assert float((b0 - b1).abs().max()) <= (5e-2 if act == "bf16" else 2e-2) * max(float(b0.abs().max()), 1e-2), k


@pytest.mark.parametrize("batch,c_in,c_out,t", [(2, 64, 256, 501), (3, 256, 200, 77), (1, 192, 128, 1000), (32, 512, 512, 501), (5, 128, 72, 8)])
def test_bf16_pointwise_products_match_a_float64_product(batch, c_in, c_out, t):
    """The three products of a 1x1 convolution on the bf16 training path (quartznet/blocks.py:169-182 with kernel_size 1), through
    train_ops' helpers, against float64 products of the SAME bf16-rounded operands:
      forward v = W u and data gradient du = W^T dv on the inference kernel's pointwise-only mode (ts_tcs_subblock_fwd) fed with
      fragments from ts_train_pack_pw_multi -- bf16 results: half an ulp (0.4 %) of each value + f32 accumulation, 6e-3 of the scale;
      weight gradient dW = sum_b dv u^T on csrc/train_gemm.hip -- f32, 2e-3 of the scale, and it ACCUMULATES onto dw.
    The rows carry NaN bit patterns in their pitch padding: nothing may leak into frames < T; partial tiles in every dimension; the
    fragment packer is checked bit for bit against plan.pack_pw_frags (both orientations)."""
    from thunder_speech_amd import plan
    from thunder_speech_amd import train_ops as T
    g = torch.Generator().manual_seed(c_in + c_out + t)
    u = torch.randn(batch, c_in, t, generator=g).bfloat16()
    w = torch.nn.Parameter((torch.randn(c_out, c_in, 1, generator=g) / c_in ** 0.5).cuda())
    dv = torch.randn(batch, c_out, t, generator=g).bfloat16()
    w2 = w.detach().view(c_out, c_in)
    wb = w2.bfloat16().double().cpu()

    def rows(x):
        b, c, tt = x.shape
        buf = torch.full((b, c, T.row_pitch(tt)), float("nan"), dtype=torch.bfloat16, device="cuda")
        buf[:, :, :tt] = x.cuda()
        return buf[:, :, :tt]
    ur, dvr = rows(u), rows(dv)
    assert T.is_act(ur) and T.is_act(dvr)
    fw, bk = T.pw_frags(w, w2)
    assert torch.equal(fw, plan.pack_pw_frags(w2)) and torch.equal(bk, plan.pack_pw_frags(w2.t().contiguous()))
    v = T._pw_fwd(ur, w, w2)
    ref = torch.einsum("oc,bct->bot", wb, u.double())
    used_tcs = c_in % 64 == 0
    assert v.dtype == torch.bfloat16 and float((v.double().cpu() - ref).abs().max()) <= 6e-3 * float(ref.abs().max())
    du, dw = T._pw_bwd(dvr, ur, w, w2)
    ref_du = torch.einsum("oc,bot->bct", wb, dv.double())
    ref_dw = torch.einsum("bot,bct->oc", dv.double(), u.double())
    assert float((du.double().cpu() - ref_du).abs().max()) <= 6e-3 * float(ref_du.abs().max())
    assert float((dw.double().cpu() - ref_dw).abs().max()) <= 2e-3 * float(ref_dw.abs().max())
    if c_out % 64 == 0 and c_in % 8 == 0:
        T._wgrad(dvr, ur, dw)                            # it ACCUMULATES
        assert float((dw.double().cpu() - 2 * ref_dw).abs().max()) <= 4e-3 * float(ref_dw.abs().max())
    # the library path gives the same numbers to rounding
    T.set_pointwise_backend("gemm_f32")
    try:
        v2 = T._pw_fwd(ur, w, w2)
        du2, dw2 = T._pw_bwd(dvr, ur, w, w2)
    finally:
        T.set_pointwise_backend("mfma")
    assert float((v2.double() - v.double()).abs().max()) <= 8e-3 * float(ref.abs().max()) or not used_tcs
    assert float((dw2.double().cpu() - ref_dw).abs().max()) <= 2e-3 * float(ref_dw.abs().max())


def test_grouped_weight_gradient_launch_equals_the_per_layer_launches():
    """ts_train_pwconv_wgrad_multi (all layers of a backward piece in ceil(n / 32) launches, workgroup -> (layer, tile) through prefix sums) followed
    by ts_train_wgrad_reduce_multi against one ts_train_pwconv_wgrad_mfma per layer: the same products over fewer, longer clip groups (1e-5).
    40 layers (two launches of the grouped kernel) of mixed shapes, one with a length mask on its input."""
    import ctypes as C
    from thunder_speech_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(4, 64, 128, 300), (3, 256, 256, 501), (5, 128, 64, 77), (2, 512, 256, 192)] * 10
    items = (_lib.WgradItem * len(shapes))()
    keep, want, got, parts = [], [], [], []
    for i, (b, ci, co, t) in enumerate(shapes):
        p = (t + 191) // 192 * 192 + 64
        dv = torch.randn(b, co, p, device="cuda", generator=g).bfloat16()
        u = torch.randn(b, ci, p, device="cuda", generator=g).bfloat16()
        lens = torch.randint(1, t + 1, (b,), device="cuda", generator=g).int() if i % 4 == 1 else None
        n_ws = L.ts_train_pwconv_wgrad_workspace(b, ci, co)
        ws1, ws2 = torch.empty(n_ws, device="cuda"), torch.empty(n_ws, device="cuda")
        dw1, dw2 = torch.zeros(co, ci, device="cuda"), torch.zeros(co, ci, device="cuda")
        assert L.ts_train_pwconv_wgrad_mfma(dv.data_ptr(), u.data_ptr(), lens.data_ptr() if lens is not None else None, dw1.data_ptr(), ws1.data_ptr(),
                                            b, ci, co, t, p, p, st) == 0
        it = items[i]
        it.dv, it.u, it.len_u, it.workspace = dv.data_ptr(), u.data_ptr(), (lens.data_ptr() if lens is not None else None), ws2.data_ptr()
        it.batch, it.c_in, it.c_out, it.t, it.pitch_u, it.pitch_v = b, ci, co, t, p, p
        keep += [dv, u, lens, ws1, ws2]
        want.append(dw1); got.append(dw2); parts.append(L.ts_train_pwconv_wgrad_multi_parts(b, ci, co))
    assert L.ts_train_pwconv_wgrad_multi(items, len(shapes), st) == 0
    n = len(shapes)
    assert L.ts_train_wgrad_reduce_multi((C.c_void_p * n)(*[keep[5 * i + 4].data_ptr() for i in range(n)]), (C.c_void_p * n)(*[d.data_ptr() for d in got]),
                                         (C.c_int64 * n)(*[d.numel() for d in got]), (C.c_int32 * n)(*parts), n, st) == 0
    torch.cuda.synchronize()
    for a, b_ in zip(want, got):                      # fewer, longer clip groups in the grouped launch: the same products, another summation order
        assert torch.allclose(a, b_, rtol=1e-5, atol=1e-5 * float(a.abs().max())) and float(a.abs().max()) > 0


def test_tile_statistics_out_of_the_pointwise_epilogue_equal_the_separate_pass():
    """ABI v8: between two repeats the BatchNorm statistics travel as per-tile (sum, sum of squares) pairs written by the 1x1 launch's epilogue
    (ts_tcs_desc.stats) and summed by the next matrix-core depthwise launch (ts_train_dwconv_fwd_bn_tiles), instead of ts_train_bn_stats reading the
    tensor back (quartznet/blocks.py:317-338 in train mode: BatchNorm over all B x T frames, quirk A4).  The pairs are formed from the f32 accumulators,
    the pass from the bf16-stored values: outputs, every gradient and the running statistics of a 3-repeat block agree to bf16 accuracy, at 24 clips
    (the matrix-core depthwise kernels need >= 17) with ragged lengths and a frame count that is no multiple of the tile."""
    from thunder_speech_amd import train_ops
    from thunder_speech_amd.quartznet.blocks import QuartznetBlock
    g = torch.Generator().manual_seed(12)
    x = torch.randn(24, 128, 333, generator=g).cuda()
    lengths = torch.randint(100, 334, (24,), generator=g).cuda()
    lengths[0] = 333
    cot = torch.randn(24, 256, 333, generator=g).cuda()
    res = {}
    train_ops.set_activation_dtype("bf16")
    try:
        for mode in (True, False):
            train_ops.TILE_STATS = mode
            torch.manual_seed(3)
            blk = QuartznetBlock(128, 256, repeat=3, kernel_size=(33,), separable=True).cuda().train()
            for p in blk.parameters():
                if p.dim() == 1:
                    p.data.add_(0.1 * torch.randn_like(p))
            y, _ = blk(x, lengths)
            yf = train_ops.from_act(y) if train_ops.is_act(y) else y
            (yf * cot).sum().backward()
            res[mode] = (yf.detach().float().clone(), {n: p.grad.clone() for n, p in blk.named_parameters()},
                         {n: b.clone() for n, b in blk.named_buffers() if "running" in n})
    finally:
        train_ops.TILE_STATS = True
        train_ops.set_activation_dtype("fp32")
    (y1, g1, r1), (y0, g0, r0) = res[True], res[False]
    assert float((y1 - y0).abs().max()) <= 3e-2 * float(y0.abs().max())
    for n in g0:            # three BatchNorm + ReLU layers deep, bf16 rounding boundaries and ReLU gates move with the last bits of the statistics
        assert float((g1[n] - g0[n]).norm()) <= 8e-2 * max(float(g0[n].norm()), 1e-6), n
    for n in r0:
        assert torch.allclose(r1[n], r0[n], rtol=2e-3, atol=2e-4), n


@pytest.mark.parametrize("b,ci,co,t", [(17, 64, 256, 77), (32, 512, 512, 501), (20, 256, 1024, 128), (24, 128, 96, 333), (33, 64, 512, 64), (18, 192, 320, 1000)])
def test_tile_statistics_of_the_pointwise_launch_sum_to_the_channel_moments(b, ci, co, t):
    """ts_tcs_desc.stats: per (channel, tile) the launch leaves (sum y, sum y^2) over the tile's frames < T, taken from its f32 accumulators; summed over the
    tiles they are the channel's moments over all B x T frames (nn.BatchNorm1d in train mode, quirk A4: padded frames count).  NaN in the rows' pitch padding
    must not leak in; wide (64-frame tiles), narrow (128 / 64) and ragged (c_out % 32 != 0, T % tile != 0) shapes."""
    from thunder_speech_amd import train_ops as T
    g = torch.Generator(device="cuda").manual_seed(5)
    p = (t + 191) // 192 * 192 + 64
    u = torch.randn(b, ci, p, device="cuda", generator=g).bfloat16()
    u[:, :, t:] = float("nan")
    w = torch.randn(co, ci, device="cuda", generator=g) / ci ** 0.5
    y = torch.empty(b, co, p, device="cuda", dtype=torch.bfloat16)
    stats = T.tile_stats_buffer(b, co, t, "cuda")
    stats.fill_(float("nan"))                                   # every entry is written by the launch
    lens = torch.full((b,), t, dtype=torch.int32, device="cuda")
    from thunder_speech_amd import plan
    T._tcs_pointwise(u[:, :, :t].as_strided((b, ci, t), (ci * p, p, 1)), plan.pack_pw_frags(w), y[:, :, :t].as_strided((b, co, t), (co * p, p, 1)), lens, co, stats)
    torch.cuda.synchronize()
    ref = torch.einsum("oc,bct->bot", w.bfloat16().double(), u[:, :, :t].double())
    got = stats.double().sum(1)
    assert torch.isfinite(stats).all()
    s1, s2 = ref.sum((0, 2)), (ref * ref).sum((0, 2))
    assert float((got[:, 0] - s1).abs().max()) <= 1e-3 * max(float(s2.sqrt().max()), 1.0)
    assert float((got[:, 1] - s2).abs().max()) <= 2e-3 * float(s2.max())
    assert float((y[:, :, :t].double() - ref).abs().max()) <= 1e-2 * float(ref.abs().max())
