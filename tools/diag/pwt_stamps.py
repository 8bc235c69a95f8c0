"""Phase timestamps of the whole-K pointwise tile kernel (variant `stamp`: -DTS_PWT_STAMP): per workgroup, 100 MHz wall clock."""
import ctypes as C, os, sys, torch, numpy as np
os.environ["TS_LIB_VARIANT"] = "stamp"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from thunder_speech_amd import _lib, plan
L = _lib.lib()
if hasattr(L, "ts_tcs_pointwise_select"): L.ts_tcs_pointwise_select(int(os.environ.get("TS_PW_TILE", "1")))     # round-6 experiment builds only
st = torch.cuda.current_stream().cuda_stream
for (b, ci, co, t) in [(32, 512, 512, 501), (32, 256, 256, 501), (5, 512, 512, 37)]:
    p = (t + 191) // 192 * 192 + 64
    u = torch.randn(b, ci, p, device="cuda").bfloat16()
    w = torch.randn(co, ci, device="cuda") / ci ** 0.5
    y = torch.empty(b, co, p, device="cuda", dtype=torch.bfloat16)
    frags, bias = plan.pack_pw_frags(w), torch.zeros((co + 31) // 32 * 32, device="cuda")
    lens = torch.full((b,), t, dtype=torch.int32, device="cuda")
    d = _lib.TcsDesc()
    d.batch, d.c_in, d.c_out, d.t_in, d.t_out, d.pitch_in, d.pitch_out = b, ci, co, t, t, p, p
    d.kernel, d.stride, d.dilation, d.padding, d.depthwise, d.relu, d.out_fp32, d.flags = 1, 1, 1, 0, 0, 0, 0, 0
    d.pw_w, d.bias = frags.data_ptr(), bias.data_ptr()
    for _ in range(5):
        assert L.ts_tcs_subblock_fwd(C.byref(d), u.data_ptr(), lens.data_ptr(), None, None, y.data_ptr(), st) == 0
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    L.ts_debug_pwt_stamps.argtypes = [C.c_void_p]
    assert L.ts_debug_pwt_stamps(buf.ctypes.data) == 0
    n = b * ((t + 63) // 64) * ((co + 511) // 512 if co > 256 else 1)
    s = buf.reshape(4096, 8)[:n, :5].astype(np.int64)
    t00 = s[:, 0].min()
    rel = (s - t00) * 10.0 / 1000.0       # us (100 MHz)
    print(f"B={b} T={t} {ci}->{co}: {n} workgroups; start spread {rel[:,0].max():.2f} us")
    for name, i, j in (("issue+wait chunk0 (start->barrier0)", 0, 1), ("chunks 0..n-2 (barrier0->last barrier)", 1, 2), ("last chunk mfma", 2, 3), ("epilogue", 3, 4), ("total", 0, 4)):
        dd = rel[:, j] - rel[:, i]
        print(f"   {name:42s} mean {dd.mean():6.2f}  min {dd.min():6.2f}  max {dd.max():6.2f} us")
    print(f"   last workgroup ends at {rel[:,4].max():.2f} us after the first started")
