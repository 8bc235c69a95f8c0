"""Time the pointwise-only mode of ts_tcs_subblock_fwd (generic kernel, flags = 0: what the bf16 training path's 1x1 forward / data gradient
launch) on the C4 shapes (32 clips x 501 frames) and check it against an f32 einsum."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from thunder_speech_amd import _lib, plan
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
if hasattr(L, "ts_tcs_pointwise_select"): L.ts_tcs_pointwise_select(int(os.environ.get("TS_PW_TILE", "1")))     # round-6 experiment builds only
shapes = [(32, 512, 512, 501), (32, 256, 256, 501), (32, 256, 512, 501), (32, 512, 256, 501), (32, 1024, 1024, 501), (32, 512, 1024, 501), (32, 1024, 512, 501),
          (256, 512, 512, 501), (256, 256, 256, 501), (64, 512, 512, 751), (5, 512, 512, 37), (3, 256, 200, 700)]
if os.environ.get("TS_PW_SHORT"):
    shapes = shapes[:4] + shapes[10:11]
for (b, ci, co, t) in shapes:
    p = (t + 191) // 192 * 192 + 64
    u = torch.randn(b, ci, p, device="cuda").bfloat16()
    w = torch.randn(co, ci, device="cuda") / ci ** 0.5
    y = torch.empty(b, co, p, device="cuda", dtype=torch.bfloat16)
    frags, bias = plan.pack_pw_frags(w), torch.zeros((co + 31) // 32 * 32, device="cuda")
    lens = torch.full((b,), t, dtype=torch.int32, device="cuda")
    if b <= 5:
        lens = torch.randint(1, t + 1, (b,), dtype=torch.int32, device="cuda")      # ragged: the input mask matters
        lens[0] = t
    d = _lib.TcsDesc()
    d.batch, d.c_in, d.c_out, d.t_in, d.t_out, d.pitch_in, d.pitch_out = b, ci, co, t, t, p, p
    d.kernel, d.stride, d.dilation, d.padding, d.depthwise, d.relu, d.out_fp32, d.flags = 1, 1, 1, 0, 0, 0, 0, 0
    d.pw_w, d.bias = frags.data_ptr(), bias.data_ptr()
    fn = lambda: L.ts_tcs_subblock_fwd(C.byref(d), u.data_ptr(), lens.data_ptr(), None, None, y.data_ptr(), st)
    for _ in range(5):
        assert fn() == 0
    torch.cuda.synchronize()
    um = u[:, :, :t].float() * (torch.arange(t, device="cuda")[None, None, :] < lens[:, None, None])
    ref = torch.einsum("oc,bct->bot", w.bfloat16().float(), um)
    err = float((y[:, :, :t].float() - ref).abs().max()) / float(ref.abs().max())
    # 50 launches replayed from a hipGraph: eager ctypes launches are host-bound below ~10 us per kernel
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        st_side = side.cuda_stream
        fn2 = lambda: L.ts_tcs_subblock_fwd(C.byref(d), u.data_ptr(), lens.data_ptr(), None, None, y.data_ptr(), st_side)
        fn2()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(50):
                fn2()
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    print(f"lib={os.environ.get('TS_LIB_VARIANT', '-'):6s} pw_tile={os.environ.get('TS_PW_TILE', '1')}  B={b:3d} T={t:4d} Cin={ci:4d} Cout={co:4d}: {us:6.1f} us  {2.0 * b * t * ci * co / us / 1e6:6.1f} TF/s  rel err {err:.1e}")
