"""A/B of the wide-frame consumer tile for tail-zero pointwise-only layers (ts_tcs_pointwise_wide): bit-equality + time per launch (hipGraph replay)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from thunder_speech_amd import _lib, plan, tensors as TS
L = _lib.lib()
torch.manual_seed(0)
for (b, ci, co, t, res) in [(32, 1024, 1024, 1001, False), (64, 512, 1024, 751, False), (32, 1024, 1024, 251, False), (3, 512, 640, 333, False)]:
    w = torch.randn(co, ci, 1) / ci ** 0.5
    bn = [torch.rand(co) + 0.5, torch.randn(co) * 0.1, torch.randn(co) * 0.1, torch.rand(co) + 0.5]
    layer = plan.make_tcs_layer("cuda", dw_w=None, pw_w=w, bn=bn, kernel=1, stride=1, dilation=1, padding=0, relu=True)
    lens = torch.randint(t // 2, t + 1, (b,), dtype=torch.int32, device="cuda"); lens[0] = t
    xb = TS.arena(("t", 0), b, ci, t, "cuda")
    xb[:, :, :t] = torch.randn(b, ci, t, device="cuda").to(torch.bfloat16)
    for i in range(b):
        xb[i, :, int(lens[i]):] = 0
    outs, times = [], []
    for wide in (0, 1, 0, 1):
        L.ts_tcs_pointwise_wide(wide)
        out = TS.arena(("o", wide), b, co, t, "cuda"); out.zero_()
        run = lambda: layer.run(xb, t, lens, out=out, in_tail_zero=True, zero_tail=True)
        run(); torch.cuda.synchronize()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(20):
                    run()
        torch.cuda.current_stream().wait_stream(side)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 100 * 1e3)
        outs.append(out.clone())
    L.ts_tcs_pointwise_wide(0)
    print(f"B={b} {ci}->{co} T={t}: 96x512 tiles {times[0]:.1f} / {times[2]:.1f} us, 192x256 wide-frame tiles {times[1]:.1f} / {times[3]:.1f} us; bit-equal: {bool(torch.equal(outs[0], outs[1]))}", flush=True)
