"""GPU parity of the matrix-core depthwise backward (csrc/train_enc.hip dw_bwd_mfma_kernel, selected by ts_train_dwconv_bwd_select) -- the backward
of the depthwise MaskedConv1d of a training-mode QuartznetBlock (/root/reference/src/thunder/quartznet/blocks.py:95-164, 317-338) on bf16 rows --
against float64 autograd through F.conv1d on the same bf16 inputs, and against the packed-f32 FIR kernel it replaces.  Tolerances: dx is stored in
bf16 (half an ulp = 0.4 % of a value; 6e-3 of the tensor's scale covers it), dw is an f32 sum of exact bf16 products (1e-5), the folded BatchNorm's
sums are taken over the bf16-rounded gradient (4e-3)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(mode, b, ch, t, k, lens, bn, relu=True, no_dw=False):
    from thunder_speech_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    old = L.ts_train_dwconv_bwd_select(mode)
    try:
        p = (k - 1) // 2
        pitch = (t + 191) // 192 * 192 + 64
        g = torch.Generator(device="cuda").manual_seed(1)
        dy = torch.randn(b, ch, pitch, device="cuda", generator=g).bfloat16()
        x = torch.randn(b, ch, pitch, device="cuda", generator=g).bfloat16()
        x[:, :, t:] = float("nan")                      # columns >= T are scratch: nothing may leak out of them
        dy[:, :, t:] = float("nan")
        w = torch.randn(ch, k, device="cuda", generator=g) / k ** 0.5
        li = torch.tensor(lens, dtype=torch.int32, device="cuda")
        dx = torch.full((b, ch, pitch), 7.0, device="cuda").bfloat16()
        dw = torch.zeros(ch, k, device="cuda")
        out = {}
        if bn:
            mr = torch.stack([0.1 * torch.randn(ch, device="cuda", generator=g), 1.0 + 0.2 * torch.rand(ch, device="cuda", generator=g)], 1).contiguous()
            gamma = 1.0 + 0.1 * torch.randn(ch, device="cuda", generator=g)
            beta = 0.1 * torch.randn(ch, device="cuda", generator=g)
            dgam, dbet = torch.zeros(ch, device="cuda"), torch.zeros(ch, device="cuda")
            rc = L.ts_train_dwconv_bwd_bn(dy.data_ptr(), x.data_ptr(), mr.data_ptr(), gamma.data_ptr(), beta.data_ptr(), int(relu), li.data_ptr(), li.data_ptr(),
                                          w.data_ptr(), dx.data_ptr(), None if no_dw else dw.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), b, ch, t, k, p, pitch, 1, st)
            out["dgamma"], out["dbeta"] = dgam, dbet
        else:
            rc = L.ts_train_dwconv_bwd(dy.data_ptr(), x.data_ptr(), li.data_ptr(), li.data_ptr(), w.data_ptr(), dx.data_ptr(), None if no_dw else dw.data_ptr(),
                                       b, ch, t, t, k, 1, 1, p, pitch, pitch, 1, st)
        assert rc == 0
        torch.cuda.synchronize()
        out["dx"], out["dw"] = dx[:, :, :t].float(), dw
        mask = (torch.arange(t, device="cuda")[None, :] < li[:, None]).double()[:, None, :]
        xd = torch.nan_to_num(x[:, :, :t].double())
        dyd = torch.nan_to_num(dy[:, :, :t].double()) * mask
        if bn:
            sc = (gamma * mr[:, 1]).double()[None, :, None]
            hs = (beta - mr[:, 0] * gamma * mr[:, 1]).double()[None, :, None]
            pre = xd * sc + hs
            yv = (torch.relu(pre) if relu else pre).bfloat16().double()
        else:
            yv = xd
        xg = (yv * mask).requires_grad_(True)
        wg = (w.bfloat16().double() if mode else w.double()).requires_grad_(True)      # the matrix-core kernel rounds the taps to bf16, like the forward
        y = torch.nn.functional.conv1d(xg, wg[:, None, :], padding=p, groups=ch)
        (y * dyd).sum().backward()
        ref = {"dx": xg.grad * mask, "dw": wg.grad}
        if bn:
            gate = ((pre > 0).double() if relu else torch.ones_like(pre)) * mask
            gref = ref["dx"] * gate
            xhat = (xd - mr[:, 0].double()[None, :, None]) * mr[:, 1].double()[None, :, None]
            ref = {"dx": gref, "dw": wg.grad, "dbeta": gref.sum((0, 2)), "dgamma": (gref * xhat).sum((0, 2))}
        return out, ref
    finally:
        L.ts_train_dwconv_bwd_select(old)


CASES = [(3, 32, 300, 33, [300, 211, 97]), (2, 16, 77, 5, [77, 40]), (5, 64, 501, 63, [501, 499, 3, 256, 257]), (2, 48, 700, 75, [700, 512]),
         (4, 32, 256, 39, [256, 255, 1, 129]), (2, 16, 520, 51, [520, 260]), (1, 16, 128, 3, [128]), (2, 16, 1030, 21, [1030, 0])]


@pytest.mark.parametrize("mode", [1, 2])                       # 128-frame tiles (two workgroups per CU) and 256-frame tiles
@pytest.mark.parametrize("bn", [False, True])
@pytest.mark.parametrize("b,ch,t,k,lens", CASES)
def test_matrix_core_depthwise_backward_matches_float64_autograd(mode, bn, b, ch, t, k, lens):
    out, ref = _run(mode, b, ch, t, k, lens, bn)
    tol = {"dx": 6e-3, "dw": 1e-5, "dbeta": 4e-3, "dgamma": 4e-3}
    for key, r in ref.items():
        scale = max(float(r.abs().max()), 1e-6)
        err = float((out[key].double() - r).abs().max()) / scale
        assert torch.isfinite(out[key]).all() and err <= tol[key], (key, err)


def test_frozen_weight_and_identity_affine_variants():
    """dw == NULL (the reference's first fine-tuning phase freezes the convolutions): data gradient only, same values; a folded BatchNorm without
    ReLU (block tails) gates nothing."""
    full, _ = _run(1, 3, 32, 300, 33, [300, 211, 97], True)
    part, _ = _run(1, 3, 32, 300, 33, [300, 211, 97], True, no_dw=True)
    assert torch.equal(full["dx"], part["dx"]) and float(part["dw"].abs().max()) == 0.0
    out, ref = _run(1, 3, 32, 300, 33, [300, 211, 97], True, relu=False)
    for key, r in ref.items():
        assert float((out[key].double() - r).abs().max()) <= 6e-3 * max(float(r.abs().max()), 1e-6), key


def test_select_falls_back_to_the_fir_kernel_for_other_geometries():
    """Channel counts that are not a multiple of 16 (and f32 rows, dilation, stride) keep the packed-f32 FIR kernel whatever the mode says."""
    out1, ref = _run(1, 2, 24, 200, 11, [200, 150], False)
    out0, _ = _run(0, 2, 24, 200, 11, [200, 150], False)
    assert torch.equal(out1["dx"], out0["dx"])
    assert float((out1["dw"].double() - _run(0, 2, 24, 200, 11, [200, 150], False)[1]["dw"]).abs().max()) <= 1e-4 * float(ref["dw"].abs().max())


def test_random_geometries_sweep():
    """60 random geometries (odd K in [3, 75], 16 / 32 / 48 channels, 8..900 frames, ragged lengths incl. empty and full clips, both tile sizes, with and
    without the folded BatchNorm): the template with more k-steps / tap groups than a K needs, tiles that end exactly at T, clips shorter than the halo."""
    import random
    rnd = random.Random(2024)
    tol = {"dx": 6e-3, "dw": 2e-5, "dbeta": 5e-3, "dgamma": 5e-3}
    for case in range(60):
        k = rnd.choice(range(3, 77, 2))
        ch = rnd.choice([16, 32, 48])
        t = rnd.choice([8, 64, 127, 128, 129, 255, 256, 257, 384, 500, 512, 640, 900]) if case % 3 else rnd.randint(8, 900)
        b = rnd.randint(1, 4)
        lens = [rnd.choice([0, 1, t, t, rnd.randint(0, t)]) for _ in range(b)]
        bn, mode = bool(case & 1), 1 + (case >> 1 & 1)
        out, ref = _run(mode, b, ch, t, k, lens, bn, relu=bool(case & 4) or not bn)
        for key, r in ref.items():
            scale = max(float(r.abs().max()), 1e-6)
            err = float((out[key].double() - r).abs().max()) / scale
            assert torch.isfinite(out[key]).all() and err <= tol[key], (case, (mode, b, ch, t, k, lens, bn), key, err)


def test_strided_depthwise_backward_without_a_data_gradient():
    """ts_train_dwconv_bwd(dx = NULL): the stem's input are the features and need no gradient (blocks.py:317-338 under module.py:102-127) -- on the strided
    geometries, whose two gradients are separate launches, only the weight gradient is formed, and it is the one the full call gives; the fused stride-1
    kernels and a call that asks for nothing refuse."""
    from thunder_speech_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    b, ch, t_in, k, stride = 3, 64, 401, 33, 2
    pad = (k - 1) // 2
    t_out = (t_in + 2 * pad - (k - 1) - 1) // stride + 1
    pin, pout = (t_in + 191) // 192 * 192 + 64, (t_out + 191) // 192 * 192 + 64
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(b, ch, pin, device="cuda", generator=g).bfloat16()
    dy = torch.randn(b, ch, pout, device="cuda", generator=g).bfloat16()
    w = torch.randn(ch, k, device="cuda", generator=g) / k ** 0.5
    li = torch.tensor([t_in, 300, 77], dtype=torch.int32, device="cuda")
    lo = ((li + 2 * pad - (k - 1) - 1) // stride + 1).to(torch.int32)
    dx = torch.zeros(b, ch, pin, device="cuda").bfloat16()
    dw_full, dw_only = torch.zeros(ch, k, device="cuda"), torch.zeros(ch, k, device="cuda")
    args = lambda dxp, dwp: (dy.data_ptr(), x.data_ptr(), li.data_ptr(), lo.data_ptr(), w.data_ptr(), dxp, dwp, b, ch, t_in, t_out, k, stride, 1, pad, pin, pout, 1, st)
    assert L.ts_train_dwconv_bwd(*args(dx.data_ptr(), dw_full.data_ptr())) == 0
    assert L.ts_train_dwconv_bwd(*args(None, dw_only.data_ptr())) == 0
    torch.cuda.synchronize()
    torch.testing.assert_close(dw_only, dw_full, rtol=1e-5, atol=1e-5)        # float atomics over clip groups: the same sum in another order
    assert float(dw_full.abs().max()) > 0
    TS_EINVAL = -1
    assert L.ts_train_dwconv_bwd(*args(None, None)) == TS_EINVAL                           # nothing asked for
    # a fused stride-1 geometry forms both gradients in one pass and wants the buffer
    t = 300
    p1 = (t + 191) // 192 * 192 + 64
    x1, dy1 = torch.randn(2, 64, p1, device="cuda", generator=g).bfloat16(), torch.randn(2, 64, p1, device="cuda", generator=g).bfloat16()
    l1 = torch.tensor([t, 200], dtype=torch.int32, device="cuda")
    rc = L.ts_train_dwconv_bwd(dy1.data_ptr(), x1.data_ptr(), l1.data_ptr(), l1.data_ptr(), w.data_ptr(), None, dw_only.data_ptr(), 2, 64, t, t, k, 1, 1, pad, p1, p1, 1, st)
    assert rc == TS_EINVAL
