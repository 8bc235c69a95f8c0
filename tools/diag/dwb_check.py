"""Matrix-core depthwise backward (ts_train_dwconv_bwd_select(1)) against the packed-f32 FIR kernel (select(0)) and an f64 reference, with and
without the folded BatchNorm, + timing on the C4 shapes."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from thunder_speech_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)

def run(mode, b, ch, t, k, lens, bn, timing=False, no_dw=False):
    L.ts_train_dwconv_bwd_select(mode)
    p = (k - 1) // 2
    pitch = (t + 191) // 192 * 192 + 64
    g = torch.Generator(device="cuda").manual_seed(1)
    dy = torch.randn(b, ch, pitch, device="cuda", generator=g).bfloat16()
    x = torch.randn(b, ch, pitch, device="cuda", generator=g).bfloat16()
    w = torch.randn(ch, k, device="cuda", generator=g) / k ** 0.5
    li = torch.tensor(lens, dtype=torch.int32, device="cuda")
    dx = torch.full((b, ch, pitch), 7.0, device="cuda").bfloat16()
    dw = torch.zeros(ch, k, device="cuda")
    if bn:
        mr = torch.stack([0.1 * torch.randn(ch, device="cuda", generator=g), 1.0 + 0.2 * torch.rand(ch, device="cuda", generator=g)], 1).contiguous()
        gamma = 1.0 + 0.1 * torch.randn(ch, device="cuda", generator=g); beta = 0.1 * torch.randn(ch, device="cuda", generator=g)
        dgam = torch.zeros(ch, device="cuda"); dbet = torch.zeros(ch, device="cuda")
        fn = lambda: L.ts_train_dwconv_bwd_bn(dy.data_ptr(), x.data_ptr(), mr.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1, li.data_ptr(), li.data_ptr(),
                                              w.data_ptr(), dx.data_ptr(), None if no_dw else dw.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), b, ch, t, k, p, pitch, 1, st)
    else:
        fn = lambda: L.ts_train_dwconv_bwd(dy.data_ptr(), x.data_ptr(), li.data_ptr(), li.data_ptr(), w.data_ptr(), dx.data_ptr(), None if no_dw else dw.data_ptr(),
                                           b, ch, t, t, k, 1, 1, p, pitch, pitch, 1, st)
    assert fn() == 0
    torch.cuda.synchronize()
    out = {"dx": dx[:, :, :t].float().clone(), "dw": dw.clone()}
    if bn:
        out["dgamma"], out["dbeta"] = dgam.clone(), dbet.clone()
    # f64 reference
    mask = (torch.arange(t, device="cuda")[None, :] < li[:, None]).double()[:, None, :]
    xd, dyd = x[:, :, :t].double(), dy[:, :, :t].double() * mask
    if bn:
        sc = (gamma * mr[:, 1]).double()[None, :, None]; hs = (beta - mr[:, 0] * gamma * mr[:, 1]).double()[None, :, None]
        yv = torch.relu(xd * sc + hs).bfloat16().double()
    else:
        yv = xd
    xm = yv * mask
    wd = w.bfloat16().double() if mode == 1 else w.double()
    xg = xm.clone().requires_grad_(True); wg = wd.clone().requires_grad_(True)
    y = torch.nn.functional.conv1d(xg, wg[:, None, :], padding=p, groups=ch)
    (y * dyd).sum().backward()
    ref_dx = xg.grad * mask
    if bn:
        gate = ((xd * sc + hs) > 0).double() * mask
        gref = ref_dx * gate
        out["ref"] = {"dx": gref, "dw": wg.grad, "dbeta": gref.sum((0, 2)), "dgamma": (gref * (xd - mr[:, 0].double()[None, :, None]) * mr[:, 1].double()[None, :, None]).sum((0, 2))}
    else:
        out["ref"] = {"dx": ref_dx, "dw": wg.grad}
    if timing:
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        out["us"] = e0.elapsed_time(e1) / 50 * 1e3
    return out

cases = [(3, 32, 300, 33, [300, 211, 97]), (2, 16, 77, 5, [77, 40]), (5, 64, 501, 63, [501, 499, 3, 256, 257]), (2, 48, 700, 75, [700, 512]), (4, 32, 256, 39, [256, 255, 1, 129]), (2, 16, 520, 51, [520, 260])]
for (b, ch, t, k, lens) in cases:
    for bn in (False, True):
        o1, o0 = run(1, b, ch, t, k, lens, bn), run(0, b, ch, t, k, lens, bn)
        o2 = run(2, b, ch, t, k, lens, bn)
        assert all(torch.allclose(o1[key], o2[key], rtol=2e-2, atol=1e-3 * float(o1["ref"][key].abs().max())) for key in o1["ref"]), "tile 128 vs 256"
        msg = []
        for key in o1["ref"]:
            ref = o1["ref"][key].float()
            e1 = float((o1[key] - ref).abs().max()) / max(float(ref.abs().max()), 1e-6)
            e0 = float((o0[key] - o0["ref"][key].float()).abs().max()) / max(float(ref.abs().max()), 1e-6)
            msg.append(f"{key}: mfma {e1:.1e} / fir {e0:.1e}")
        print(f"B={b} C={ch} T={t} K={k} bn={int(bn)}  " + "  ".join(msg))
for (b, ch, t, k) in [(32, 512, 501, 63), (32, 512, 501, 75), (32, 512, 501, 51), (32, 256, 501, 33), (32, 256, 501, 39)]:
    for bn in (False, True):
        o1, o0 = run(1, b, ch, t, k, [t] * b, bn, True), run(0, b, ch, t, k, [t] * b, bn, True)
        n1, n0 = run(1, b, ch, t, k, [t] * b, bn, True, True), run(0, b, ch, t, k, [t] * b, bn, True, True)
        o2, n2 = run(2, b, ch, t, k, [t] * b, bn, True), run(2, b, ch, t, k, [t] * b, bn, True, True)
        print(f"B={b} C={ch} T={t} K={k} bn={int(bn)}: mfma128 {o1['us']:6.1f} us  mfma256 {o2['us']:6.1f}  fir {o0['us']:6.1f} us   | data gradient only: mfma128 {n1['us']:6.1f}  mfma256 {n2['us']:6.1f}  fir {n0['us']:6.1f}")
