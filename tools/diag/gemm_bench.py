"""Time the training path's pointwise GEMM entry points (own MFMA kernels vs the rocBLAS ones) on one layer shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from thunder_speech_amd import _lib
L = _lib.lib()
shapes = [(32, 512, 512, 501), (32, 256, 256, 501), (32, 1024, 1024, 501), (32, 256, 512, 501), (32, 512, 256, 501), (32, 512, 1024, 501), (32, 1024, 512, 501)] if len(sys.argv) < 2 else [tuple(int(v) for v in sys.argv[1].split(","))]
st = torch.cuda.current_stream().cuda_stream
for (b, ci, co, t) in shapes:
    p = (t + 191) // 192 * 192 + 64
    u = torch.randn(b, ci, p, device="cuda").bfloat16()
    dv = torch.randn(b, co, p, device="cuda").bfloat16()
    w = (torch.randn(co, ci, device="cuda") / ci ** 0.5).bfloat16()
    v = torch.empty(b, co, p, device="cuda", dtype=torch.bfloat16)
    du = torch.empty(b, ci, p, device="cuda", dtype=torch.bfloat16)
    dw = torch.zeros(co, ci, device="cuda")
    ws = torch.empty(b * co * ci, device="cuda")
    stats = torch.zeros(co, 2, device="cuda", dtype=torch.float64)
    wsw = torch.empty(L.ts_train_pwconv_wgrad_workspace(b, ci, co), device="cuda")
    calls = {
        "fwd_rocblas": lambda: L.ts_train_pwconv_fwd(u.data_ptr(), w.data_ptr(), v.data_ptr(), b, ci, co, t, p, p, 2, st),
        "wgrad_mfma": lambda: L.ts_train_pwconv_wgrad_mfma(dv.data_ptr(), u.data_ptr(), dw.data_ptr(), wsw.data_ptr(), b, ci, co, t, p, p, st),
        "bwd_rocblas(du+dw)": lambda: L.ts_train_pwconv_bwd(dv.data_ptr(), u.data_ptr(), w.data_ptr(), du.data_ptr(), dw.data_ptr(), ws.data_ptr(), b, ci, co, t, p, p, 2, st),
    }
    flops = 2.0 * b * t * ci * co
    print(f"shape B={b} Cin={ci} Cout={co} T={t}: {flops/1e9:.2f} GFLOP per product")
    for name, fn in calls.items():
        for _ in range(5):
            rc = fn()
        assert rc == 0, (name, rc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        k = 2 if "du+dw" in name else 1
        print(f"  {name:22s} {us:8.1f} us   {k * flops / us / 1e6:7.1f} TF/s")

# ---- the inference kernel's pointwise-only mode (split kernel, identity stages) on the same product
import ctypes as C
from thunder_speech_amd import plan
for (b, ci, co, t) in shapes:
    p = (t + 191) // 192 * 192 + 64
    u = torch.randn(b, ci, p, device="cuda").bfloat16()
    w = (torch.randn(co, ci, device="cuda") / ci ** 0.5)
    y = torch.empty(b, co, p, device="cuda", dtype=torch.bfloat16)
    frags = plan.pack_pw_frags(w)
    bias = torch.zeros((co + 31) // 32 * 32, device="cuda")
    lens = torch.full((b,), t, dtype=torch.int32, device="cuda")
    d = _lib.TcsDesc()
    d.batch, d.c_in, d.c_out, d.t_in, d.t_out, d.pitch_in, d.pitch_out = b, ci, co, t, t, p, p
    d.kernel, d.stride, d.dilation, d.padding, d.depthwise, d.relu, d.out_fp32 = 1, 1, 1, 0, 0, 0, 0
    for flags in (1, 0):
        d.flags = flags
        d.pw_w, d.bias = frags.data_ptr(), bias.data_ptr()
        fn = lambda: L.ts_tcs_subblock_fwd(C.byref(d), u.data_ptr(), lens.data_ptr(), None, None, y.data_ptr(), st)
        for _ in range(5):
            rc = fn()
        assert rc == 0, rc
        torch.cuda.synchronize()
        ref = torch.einsum("oc,bct->bot", w.bfloat16().float(), u[:, :, :t].float())
        err = float((y[:, :, :t].float() - ref).abs().max()) / float(ref.abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"tcs pointwise-only flags={flags} B={b} Cin={ci} Cout={co} T={t} pitch {p}: {us:8.1f} us  {2.0*b*t*ci*co/us/1e6:7.1f} TF/s  rel err {err:.2e}")
