"""Time the decoder launch (pointwise 1024 -> 29, fp32 logits, generic kernel) at the C2 size and check it against an f32 einsum."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from thunder_speech_amd import _lib, plan
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
for (b, ci, co, t) in [(64, 1024, 29, 751), (32, 1024, 29, 501), (3, 640, 129, 77)]:
    p = _lib.time_pitch(t)
    u = torch.randn(b, ci, p, device="cuda").bfloat16()
    w = torch.randn(co, ci, device="cuda") / ci ** 0.5
    y = torch.empty(b, co, p, device="cuda", dtype=torch.float32)
    frags, bias = plan.pack_pw_frags(w), torch.randn((co + 31) // 32 * 32, device="cuda")
    lens = torch.tensor([t - 7 * i for i in range(b)], dtype=torch.int32, device="cuda").clamp(min=1)
    d = _lib.TcsDesc()
    d.batch, d.c_in, d.c_out, d.t_in, d.t_out, d.pitch_in, d.pitch_out = b, ci, co, t, t, p, p
    d.kernel, d.stride, d.dilation, d.padding, d.depthwise, d.relu, d.out_fp32, d.flags = 1, 1, 1, 0, 0, 0, 1, 0
    d.pw_w, d.bias = frags.data_ptr(), bias.data_ptr()
    fn = lambda: L.ts_tcs_subblock_fwd(C.byref(d), u.data_ptr(), lens.data_ptr(), None, None, y.data_ptr(), st)
    for _ in range(5):
        assert fn() == 0
    torch.cuda.synchronize()
    mask = (torch.arange(t, device="cuda")[None, :] < lens[:, None]).float()[:, None, :]
    ref = torch.einsum("oc,bct->bot", w.bfloat16().float(), u[:, :, :t].float() * mask) + bias[:co][None, :, None]
    err = float((y[:, :, :t] - ref).abs().max()) / float(ref.abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"decoder B={b} Cin={ci} Cout={co} T={t}: {e0.elapsed_time(e1) / 100 * 1e3:6.1f} us   rel err {err:.1e}")
