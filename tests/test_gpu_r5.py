"""Round-5 GPU tests, all through the C ABI: the squeeze-excite bottleneck kernels (forward and autograd backward) against float64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ptr(t):
    return t.data_ptr()


@pytest.mark.parametrize("b,c,r", [(32, 1024, 128), (3, 24, 3), (5, 200, 25), (1, 64, 8), (7, 1536, 192)])
def test_squeeze_excite_bottleneck_forward_and_backward_match_float64_autograd(b, c, r):
    """ts_train_se_gate_fwd / _bwd (citrinet/blocks.py:72-83: Linear -> ReLU -> Linear -> sigmoid, both bias-free) vs torch float64 autograd;
    the inference entry ts_se_gate_fwd keeps its own two launches (faster at 32 x 1024)."""
    from thunder_speech_amd import _lib
    L = _lib.lib()
    gen = torch.Generator().manual_seed(b * 1000 + c)
    mean = torch.randn(b, c, generator=gen)
    w1 = torch.randn(r, c, generator=gen) / c ** 0.5
    w2 = torch.randn(c, r, generator=gen) / r ** 0.5
    dg = torch.randn(b, c, generator=gen)
    m64, a64, b64 = mean.double().requires_grad_(True), w1.double().requires_grad_(True), w2.double().requires_grad_(True)
    h64 = torch.relu(m64 @ a64.t())
    g64 = torch.sigmoid(h64 @ b64.t())
    (g64 * dg.double()).sum().backward()

    d = lambda t: t.to(DEV).contiguous()
    mean_d, w1_d, w2_d, dg_d = d(mean), d(w1), d(w2), d(dg)
    f32 = dict(dtype=torch.float32, device=DEV)
    h, g = torch.full((b, r), float("nan"), **f32), torch.full((b, c), float("nan"), **f32)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.ts_train_se_gate_fwd(_ptr(mean_d), _ptr(w1_d), _ptr(w2_d), _ptr(h), _ptr(g), b, c, r, st), "fwd")
    np.testing.assert_allclose(h.cpu().numpy(), h64.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(g.cpu().numpy(), g64.detach().numpy(), atol=2e-5)
    g2 = torch.full((b, c), float("nan"), **f32)                  # hid == NULL (the inference call)
    _lib.check(L.ts_train_se_gate_fwd(_ptr(mean_d), _ptr(w1_d), _ptr(w2_d), None, _ptr(g2), b, c, r, st), "fwd")
    assert torch.equal(g, g2)

    nan = lambda *s: torch.full(s, float("nan"), **f32)
    dz, dh, dmean, dw1, dw2 = nan(b, c), nan(b, r), nan(b, c), nan(r, c), nan(c, r)
    _lib.check(L.ts_train_se_gate_bwd(_ptr(dg_d), _ptr(g), _ptr(h), _ptr(mean_d), _ptr(w1_d), _ptr(w2_d), _ptr(dz), _ptr(dh), _ptr(dmean),
                                      _ptr(dw1), _ptr(dw2), b, c, r, st), "bwd")
    for got, want in ((dmean, m64.grad), (dw1, a64.grad), (dw2, b64.grad)):
        s = max(float(want.abs().max()), 1e-6)
        assert float((got.cpu().double() - want).abs().max()) <= 2e-5 * s + 1e-7


def test_squeeze_excite_train_node_uses_no_aten_matmul():
    """SqueezeExciteTrain (train_ops.py) end to end vs float64 autograd, with torch's matmul poisoned: the bottleneck runs on the library's kernels."""
    from thunder_speech_amd import train_ops as T
    gen = torch.Generator().manual_seed(3)
    b, c, r, t = 4, 64, 8, 53
    x = torch.randn(b, c, t, generator=gen)
    w1, w2 = torch.randn(r, c, generator=gen) / 8, torch.randn(c, r, generator=gen) / 3
    cot = torch.randn(b, c, t, generator=gen)
    x64, a64, b64 = x.double().requires_grad_(True), w1.double().requires_grad_(True), w2.double().requires_grad_(True)
    gate = torch.sigmoid(torch.relu(x64.mean(-1) @ a64.t()) @ b64.t())
    ((x64 * gate[:, :, None]) * cot.double()).sum().backward()

    xd = x.to(DEV).requires_grad_(True)
    w1d, w2d = w1.to(DEV).requires_grad_(True), w2.to(DEV).requires_grad_(True)
    real = torch.Tensor.__matmul__
    try:
        torch.Tensor.__matmul__ = lambda *a: (_ for _ in ()).throw(AssertionError("ATen matmul on the squeeze-excite path"))
        y = T.SqueezeExciteTrain.apply(xd, w1d, w2d)
        (y.float() * cot.to(DEV)).sum().backward()
    finally:
        torch.Tensor.__matmul__ = real
    for got, want in ((xd.grad, x64.grad), (w1d.grad, a64.grad), (w2d.grad, b64.grad)):
        s = max(float(want.abs().max()), 1e-6)
        assert float((got.cpu().double() - want).abs().max()) <= 1e-4 * s


@pytest.mark.parametrize("act,b,c,t", [("bf16", 32, 40, 501), ("bf16", 5, 24, 37), ("bf16", 8, 16, 1300), ("fp32", 16, 24, 501), ("fp32", 3, 8, 700)])
def test_block_tail_in_one_launch_each_way_matches_float64_autograd(act, b, c, t):
    """ts_train_bn2_add_relu_chan_fwd / ts_train_bn2_chan_bwd (a workgroup per channel, rows in registers): out = relu(BN(va) + BN(vb)) with batch
    statistics over all B * T frames (quartznet/blocks.py:332-337 in train mode), running-statistics update, and both BatchNorm backwards -- against
    float64 autograd; NaN in the row padding must not leak."""
    from thunder_speech_amd import _lib, train_ops as T
    L = _lib.lib()
    dt = torch.bfloat16 if act == "bf16" else torch.float32
    gen = torch.Generator().manual_seed(b * 100 + c)
    mk = lambda scale, shift: (scale * torch.randn(b, c, t, generator=gen) + shift).to(dt).float()
    va, vb, dout = mk(2.0, 0.5), mk(0.7, -1.0), mk(1.0, 0.0)
    ga, ba, gb, bb = (torch.randn(c, generator=gen) for _ in range(4))
    eps = 1e-3
    ref_in = [x.double().requires_grad_(True) for x in (va, vb, ga, ba, gb, bb)]

    def bn(v, g, be):
        mu = v.mean(dim=(0, 2), keepdim=True)
        var = v.var(dim=(0, 2), unbiased=False, keepdim=True)
        return (v - mu) / torch.sqrt(var + eps) * g[None, :, None] + be[None, :, None], mu.flatten(), var.flatten()

    ya, mua, vara = bn(ref_in[0], ref_in[2], ref_in[3])
    yb, mub, varb = bn(ref_in[1], ref_in[4], ref_in[5])
    out64 = torch.relu(ya + yb)
    (out64 * dout.double()).sum().backward()

    def rows(x):
        r = T.alloc(b, c, t, DEV, dt)
        base = r.as_strided((b, c, r.stride(1)), r.stride())
        base.fill_(float("nan"))
        r.copy_(x.to(DEV))
        return r

    va_d, vb_d, dout_d = rows(va), rows(vb), rows(dout)
    out_d = T.alloc(b, c, t, DEV, dt)
    f = lambda x: x.to(DEV).float().contiguous()
    ga_d, ba_d, gb_d, bb_d = f(ga), f(ba), f(gb), f(bb)
    mra, mrb = torch.empty(c, 2, device=DEV), torch.empty(c, 2, device=DEV)
    rma, rva, rmb, rvb = torch.zeros(c, device=DEV), torch.ones(c, device=DEV), torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    code, pitch = int(dt == torch.bfloat16), va_d.stride(1)
    rc = L.ts_train_bn2_add_relu_chan_fwd(_ptr(va_d), _ptr(ga_d), _ptr(ba_d), eps, _ptr(mra), _ptr(rma), _ptr(rva), 0.1, _ptr(nbt),
                                          _ptr(vb_d), _ptr(gb_d), _ptr(bb_d), eps, _ptr(mrb), _ptr(rmb), _ptr(rvb), 0.1, None, _ptr(out_d), b, c, t, pitch, code, st)
    units = b * ((t + 511) // 512)
    if units > (32 if act == "bf16" else 16):
        assert rc == _lib.TS_EUNSUPPORTED
        return
    _lib.check(rc, "fwd")
    tol = 2e-2 if act == "bf16" else 2e-5
    scale = float(out64.abs().max())
    assert float((out_d.float().cpu().double() - out64.detach()).abs().max()) <= tol * scale
    n = b * t
    np.testing.assert_allclose(mra[:, 0].cpu().numpy(), mua.detach().numpy(), atol=1e-5)
    np.testing.assert_allclose(mrb[:, 1].cpu().numpy(), (1.0 / torch.sqrt(varb + eps)).detach().numpy(), rtol=1e-5)
    np.testing.assert_allclose(rva.cpu().numpy(), (0.9 + 0.1 * vara * n / (n - 1)).detach().numpy(), rtol=1e-5)
    np.testing.assert_allclose(rmb.cpu().numpy(), (0.1 * mub).detach().numpy(), atol=1e-6)
    assert int(nbt) == 1
    # backward with the gate of the DEVICE output (the two paths may disagree on out > 0 where the sum is a rounding error away from 0)
    dva, dvb = T.alloc(b, c, t, DEV, dt), T.alloc(b, c, t, DEV, dt)
    dga, dba, dgb, dbb = (torch.full((c,), float("nan"), device=DEV) for _ in range(4))
    out_ref = rows(out64.detach().float().to(dt).float())
    _lib.check(L.ts_train_bn2_chan_bwd(_ptr(dout_d), None, None, _ptr(out_ref), _ptr(va_d), _ptr(vb_d), _ptr(ga_d), _ptr(mra), _ptr(gb_d), _ptr(mrb), _ptr(dva), _ptr(dvb),
                                       _ptr(dga), _ptr(dba), _ptr(dgb), _ptr(dbb), b, c, t, pitch, code, st), "bwd")
    for got, want in ((dva, ref_in[0].grad), (dvb, ref_in[1].grad), (dga, ref_in[2].grad), (dba, ref_in[3].grad), (dgb, ref_in[4].grad), (dbb, ref_in[5].grad)):
        s_ = max(float(want.abs().max()), 1e-6)
        assert torch.isfinite(got.float()).all()
        assert float((got.float().cpu().double() - want).abs().max()) <= (2e-2 if act == "bf16" else 1e-4) * s_
    # the same gradient arriving as TWO tensors (the block output fed the next block's main and residual branch): dout = d1 + mask(d2, len2) is
    # formed inside the kernel -- equal to the one-tensor call on the pre-added sum
    lens2 = torch.tensor([t - (7 * i) % max(t // 3, 1) for i in range(b)], dtype=torch.int32)
    d2 = mk(1.0, 0.0)
    keep = (torch.arange(t)[None, None, :] < lens2[:, None, None]).float()
    summed = rows((dout + d2 * keep).to(dt).float())
    d2_d = rows(d2)
    outs = []
    for first, second, ln in ((summed, None, None), (dout_d, d2_d, lens2.to(DEV))):
        o = [T.alloc(b, c, t, DEV, dt), T.alloc(b, c, t, DEV, dt)] + [torch.empty(c, device=DEV) for _ in range(4)]
        _lib.check(L.ts_train_bn2_chan_bwd(_ptr(first), _ptr(second) if second is not None else None, _ptr(ln) if ln is not None else None, _ptr(out_ref),
                                           _ptr(va_d), _ptr(vb_d), _ptr(ga_d), _ptr(mra), _ptr(gb_d), _ptr(mrb), _ptr(o[0]), _ptr(o[1]), _ptr(o[2]), _ptr(o[3]),
                                           _ptr(o[4]), _ptr(o[5]), b, c, t, pitch, code, st), "bwd2")
        outs.append(o)
    for x1, x2 in zip(*outs):
        assert torch.equal(x1, x2) if act == "bf16" else float((x1 - x2).abs().max()) <= 1e-5 * max(float(x1.abs().max()), 1e-6)


@pytest.mark.parametrize("b,c_in,c_out,t,relu,ragged", [(64, 1024, 29, 751, False, False), (3, 256, 11, 37, True, True), (5, 128, 32, 300, False, True),
                                                         (2, 640, 1025 % 33, 129, False, True), (4, 1024, 29, 128, False, False)])
def test_logits_kernel_matches_float64(b, c_in, c_out, t, relu, ragged):
    """csrc/pw_logits.hip behind ts_tcs_subblock_fwd(depthwise = 0, out_fp32 = 1, c_out <= 32): the CTC decoders' 1x1 conv + bias with f32 logits
    (reference blocks.py:199-216) -- vs a float64 product of the same bf16 operands; input frames >= len count as 0, NaN in the row padding beyond the
    last tile must not matter; c_in = 640 is not a multiple of 128 and takes the generic kernel (same answer)."""
    from thunder_speech_amd import _lib, plan
    gen = torch.Generator().manual_seed(b * 7 + t)
    x = torch.randn(b, c_in, t, generator=gen).to(torch.bfloat16)
    w = (torch.randn(c_out, c_in, generator=gen) / c_in ** 0.5)
    bias = torch.randn(c_out, generator=gen)
    lens = torch.tensor([t - (5 * i) % max(t // 2, 1) if ragged else t for i in range(b)], dtype=torch.int32)
    layer = plan.make_tcs_layer(torch.device(DEV), dw_w=None, pw_w=w, bn=None, kernel=1, stride=1, dilation=1, padding=0, relu=relu, bias_extra=bias,
                                out_fp32=True)
    pitch = _lib.time_pitch(t)
    xb = torch.full((b, c_in, pitch), float("nan"), dtype=torch.bfloat16, device=DEV)
    xb[:, :, :t] = x.to(DEV)
    n_tt = (t + 127) // 128
    xb[:, :, t:min(n_tt * 128, pitch)] = 0                      # what the tile may read beyond t stays finite (the library's buffers hold zeros there)
    y, t_out = layer.run(xb, t, lens.to(DEV))
    assert t_out == t and y.dtype == torch.float32
    mask = (torch.arange(t)[None, :] < lens[:, None]).double()
    ref = torch.einsum("vc,bct->bvt", w.to(torch.bfloat16).double(), x.double() * mask[:, None, :]) + bias.double()[None, :, None]
    if relu:
        ref = ref.clamp_min(0)
    got = y[:, :, :t].cpu().double()
    assert float((got - ref).abs().max()) <= 2e-4 * max(float(ref.abs().max()), 1.0)
