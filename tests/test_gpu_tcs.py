"""GPU parity: fused TCS sub-block kernel (through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import tcs as otcs
from oracle.primitives import bf16_round, same_padding

pytestmark = pytest.mark.gpu


def _pack(x, pitch):
    b, c, t = x.shape
    xp = torch.zeros(b, c, pitch, dtype=torch.bfloat16, device="cuda")
    xp[:, :, :t] = x.to("cuda").to(torch.bfloat16)
    # poison the pitch padding: the kernel must never let it leak into valid outputs
    xp[:, :, t:] = 7.0
    return xp


def _run_case(cin, cout, k, stride, dil, t, lens, residual, separable=True, seed=0, relu=True):
    from thunder_speech_amd import _lib, plan
    spec = otcs.BlockSpec(cin, cout, repeat=1, kernel=k, stride=stride, dilation=dil, residual=residual,
                          separable=separable)
    sd = otcs.synth_encoder_state([spec], seed=seed)
    sd = {key[2:]: v for key, v in sd.items()}
    g = torch.Generator().manual_seed(seed)
    x = bf16_round(torch.randn(len(lens), cin, t, generator=g))
    lengths = torch.tensor(lens)
    ref, ref_len = otcs.block_forward(spec, sd, "", x, lengths, emulate_bf16=True)
    ref32, _ = otcs.block_forward(spec, sd, "", x, lengths, emulate_bf16=False)

    pad = same_padding(k, stride, dil)
    bnk = "mconv.2.layer.0." if separable else "mconv.1.layer.0."
    bn = [sd[bnk + n] for n in ("weight", "bias", "running_mean", "running_var")]
    kw = dict(dw_w=sd["mconv.0.conv.weight"] if separable else None,
              pw_w=sd["mconv.1.conv.weight"] if separable else sd["mconv.0.conv.weight"], bn=bn,
              kernel=k, stride=stride, dilation=dil, padding=pad, relu=relu)
    if residual:
        kw.update(res_w=sd["res.0.conv.weight"], res_stride=spec.residual_stride,
                  res_bn=[sd["res.1.layer.0." + n] for n in ("weight", "bias", "running_mean", "running_var")])
    layer = plan.make_tcs_layer("cuda", **kw)
    xp = _pack(x, _lib.time_pitch(t))
    li = lengths.to(torch.int32).cuda()
    y, t_out = layer.run(xp, t, li, x_res=xp if residual else None, t_res=t, len_res=li if residual else None)
    torch.cuda.synchronize()
    assert t_out == ref.shape[-1]
    got = y[:, :, :t_out].float().cpu()
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    err32 = float((got - ref32).abs().max())
    # bf16 storage: 1 ulp = 2^-8 relative; allow a couple of ulps of the largest magnitude
    assert err <= 0.012 * scale, f"HIP vs bf16-emulating oracle: {err} (scale {scale})"
    assert err32 <= 0.04 * scale, f"HIP vs fp32 oracle: {err32} (scale {scale})"
    return got, ref


@pytest.mark.parametrize("cin,cout,k,stride,dil,t,lens,res", [
    (64, 64, 5, 1, 1, 128, [128, 100], False),
    (16, 32, 11, 1, 1, 50, [50, 33, 7], True),
    (256, 256, 33, 1, 1, 751, [751, 600, 13], True),
    (256, 512, 51, 1, 1, 300, [300, 211], True),
    (512, 512, 75, 1, 1, 200, [200, 1], False),
    (64, 256, 33, 2, 1, 1501, [1501, 1000], False),
    (512, 512, 87, 1, 2, 260, [260, 129], False),
    (80, 256, 5, 1, 1, 70, [70, 64], False),
    (24, 40, 13, 1, 3, 45, [45, 30], False),
    (32, 32, 9, 2, 1, 51, [51, 30, 10], False),
    (16, 16, 39, 1, 1, 140, [140, 139, 1], True),
    (48, 640, 41, 1, 1, 90, [90, 45], False),
])
def test_separable_subblock_matches_oracle(cin, cout, k, stride, dil, t, lens, res):
    _run_case(cin, cout, k, stride, dil, t, lens, res)


@pytest.mark.parametrize("cin,cout,t,lens", [(512, 1024, 300, [300, 150]), (24, 48, 40, [40, 21, 3]), (64, 29, 77, [77, 50])])
def test_pointwise_only_block_matches_oracle(cin, cout, t, lens):
    _run_case(cin, cout, 1, 1, 1, t, lens, False, separable=False)


def _run_case_tail_zero(cin, cout, k, stride, dil, t, lens, residual, seed=0):
    """Same comparison through the mask-free kernels: tail-zero inputs (arena, guards) and zeroed output tails."""
    from thunder_speech_amd import plan, tensors as TS
    spec = otcs.BlockSpec(cin, cout, repeat=1, kernel=k, stride=stride, dilation=dil, residual=residual)
    sd = {key[2:]: v for key, v in otcs.synth_encoder_state([spec], seed=seed).items()}
    g = torch.Generator().manual_seed(seed)
    x = bf16_round(torch.randn(len(lens), cin, t, generator=g))
    lengths = torch.tensor(lens)
    ref, ref_len = otcs.block_forward(spec, sd, "", x, lengths, emulate_bf16=True)
    pad = same_padding(k, stride, dil)
    bn = [sd["mconv.2.layer.0." + n] for n in ("weight", "bias", "running_mean", "running_var")]
    kw = dict(dw_w=sd["mconv.0.conv.weight"], pw_w=sd["mconv.1.conv.weight"], bn=bn, kernel=k, stride=stride,
              dilation=dil, padding=pad, relu=True)
    if residual:
        kw.update(res_w=sd["res.0.conv.weight"], res_stride=spec.residual_stride,
                  res_bn=[sd["res.1.layer.0." + n] for n in ("weight", "bias", "running_mean", "running_var")])
    layer = plan.make_tcs_layer("cuda", **kw)
    li = lengths.to(torch.int32).cuda()
    xi = TS.pack(x.cuda(), li, slot=("t", seed))
    assert TS.is_tail_zero(xi)
    xb = TS.backing(xi)
    t_out = layer.out_size(t)
    out = TS.arena(("to", seed), len(lens), cout, t_out, "cuda")
    out.fill_(3.0)                                     # stale data from an earlier use must be overwritten
    y, _ = layer.run(xb, t, li, x_res=xb if residual else None, t_res=t, len_res=li if residual else None, out=out,
                     in_tail_zero=True, zero_tail=True)
    torch.cuda.synchronize()
    got = y[:, :, :t_out].float().cpu()
    scale = max(1.0, float(ref.abs().max()))
    for b, n in enumerate(ref_len.tolist()):
        n = int(n)
        assert float((got[b, :, :n] - ref[b, :, :n]).abs().max()) <= 0.012 * scale
        assert float(got[b, :, n:].abs().max()) == 0.0 if n < t_out else True      # tail-zero invariant on the output


@pytest.mark.parametrize("cin,cout,k,stride,dil,t,lens,res", [
    (256, 256, 33, 1, 1, 751, [751, 600, 13], True),
    (256, 256, 39, 1, 1, 300, [300, 299], False),
    (256, 512, 51, 1, 1, 300, [300, 211], True),
    (512, 512, 75, 1, 1, 200, [200, 1], True),
    (512, 512, 63, 1, 1, 751, [751, 640], False),
    (64, 256, 33, 2, 1, 1501, [1501, 1000], False),
    (512, 512, 87, 1, 2, 260, [260, 129], False),      # dilation 2: phase-split kernel (even / odd frames)
    (512, 512, 87, 1, 2, 751, [751, 750, 377, 2], False),
    (64, 512, 87, 1, 2, 97, [97, 96, 1], False),
    (64, 64, 5, 1, 1, 128, [128, 100], False),
    (320, 384, 11, 1, 1, 251, [251, 97], False),        # Citrinet kernel sizes: 2 / 3 / 4 passes, 96-frame tiles
    (384, 640, 25, 1, 1, 200, [200, 155], False),
    (128, 1024, 39, 1, 1, 150, [150, 64], False),       # two output-channel splits per tile
    (128, 128, 17, 1, 1, 400, [400, 201], False),       # 192-frame tiles, 2 passes
    (256, 256, 39, 1, 1, 401, [401, 333, 250], True),   # 4 passes + residual stages, several tiles per workgroup
    (256, 256, 33, 1, 1, 1400, [1400, 1399, 700, 5] * 40, True),   # more tiles than CUs: the persistent loop wraps
    (512, 512, 51, 1, 1, 570, [570, 569, 300] * 14 + [33], True),  # 258 tiles: ragged last XCD range of the tile order
    (320, 512, 25, 1, 1, 300, [300, 150, 7], True),                # 5 residual stages (odd): one-ahead identity rows
    (320, 512, 25, 1, 1, 300, [300, 150, 7] * 43 + [299], True),   # ... and several tiles per workgroup: 5 + 5 stages per tile (row-buffer parity)
    (512, 512, 63, 1, 1, 751, [751, 700] * 40, False),             # the headline shape with 640 tiles: rows fetched by the consumer waves across tiles
])
def test_tail_zero_fast_kernels_match_oracle(cin, cout, k, stride, dil, t, lens, res):
    _run_case_tail_zero(cin, cout, k, stride, dil, t, lens, res)


# (the poison test lives in tests/test_gpu_configs.py::test_padding_region_never_leaks: different garbage per run)


@pytest.mark.parametrize("k", [63, 75, 39, 87])
def test_back_to_back_launches_never_use_the_next_stages_taps(k):
    """Regression: with the tap fragments L2-hot (the same layer launched again and again) a tap DMA that refills a slot
    for the NEXT stage used to be able to land before the running stage had read the slot.  Identity pointwise, distinct
    taps per channel: any stage that sees another stage's taps is off by O(1)."""
    from thunder_speech_amd import plan, tensors as TS
    c, t, b = 512 if k != 39 else 256, 751, 2
    dil = 2 if k == 87 else 1                          # K87 is QuartzNet's dilation-2 layer (phase-split kernel)
    pad = dil * (k - 1) // 2
    g = torch.Generator().manual_seed(k)
    dw = bf16_round(torch.randn(c, 1, k, generator=g) * 0.2)
    bn = [torch.ones(c), torch.zeros(c), torch.zeros(c), torch.ones(c) - 1e-3]      # scale 1, shift 0 after folding
    layer = plan.make_tcs_layer("cuda", dw_w=dw, pw_w=torch.eye(c).reshape(c, c, 1), bn=bn, kernel=k, stride=1,
                                dilation=dil, padding=pad, relu=False)
    x = bf16_round(torch.randn(b, c, t, generator=g))
    ref = torch.nn.functional.conv1d(x.double(), dw.double(), padding=pad, dilation=dil, groups=c).float()
    li = torch.full((b,), t, dtype=torch.int32, device="cuda")
    xb = TS.backing(TS.pack(x.cuda(), li, slot=("b2b", k)))
    out = TS.arena(("b2bo", k), b, c, t, "cuda")
    scale = float(ref.abs().max())
    for it in range(60):
        y, _ = layer.run(xb, t, li, out=out, in_tail_zero=True, zero_tail=True)
        err = float((y[:, :, :t].float().cpu() - ref).abs().max())
        assert err <= 0.012 * scale, f"launch {it}: max err {err} (scale {scale})"


@pytest.mark.parametrize("cin,cout,t,lens", [(512, 1024, 751, [751, 400, 9]), (128, 256, 300, [300, 150])])
def test_pointwise_only_tail_zero_input_keeps_reference_values_beyond_length(cin, cout, t, lens):
    """The encoder's last block is caller-visible: its output is NOT tail-zeroed (quirk A2: predict() decodes all frames),
    but its input is -- the mask-free kernel must still reproduce the reference on every frame."""
    from thunder_speech_amd import plan, tensors as TS
    spec = otcs.BlockSpec(cin, cout, repeat=1, kernel=1, stride=1, dilation=1, residual=False, separable=False)
    sd = {key[2:]: v for key, v in otcs.synth_encoder_state([spec], seed=5).items()}
    g = torch.Generator().manual_seed(5)
    x = bf16_round(torch.randn(len(lens), cin, t, generator=g))
    lengths = torch.tensor(lens)
    ref, _ = otcs.block_forward(spec, sd, "", x, lengths, emulate_bf16=True)
    bn = [sd["mconv.1.layer.0." + n] for n in ("weight", "bias", "running_mean", "running_var")]
    layer = plan.make_tcs_layer("cuda", dw_w=None, pw_w=sd["mconv.0.conv.weight"], bn=bn, kernel=1, stride=1, dilation=1,
                                padding=0, relu=True)
    li = lengths.to(torch.int32).cuda()
    xb = TS.backing(TS.pack(x.cuda(), li, slot=("pwtz", cin)))
    y, t_out = layer.run(xb, t, li, in_tail_zero=True, zero_tail=False)
    torch.cuda.synchronize()
    got = y[:, :, :t_out].float().cpu()
    scale = max(1.0, float(ref.abs().max()))
    assert float((got - ref).abs().max()) <= 0.012 * scale
    assert float(got[1, :, lens[1]:].abs().max()) > 0          # relu(shift) beyond the length, as in the reference
