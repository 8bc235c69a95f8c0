"""Time the training path's depthwise entry points on one layer shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from thunder_speech_amd import _lib
from thunder_speech_amd import train_ops as T
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
for (b, c, t, k) in [(32, 512, 501, 63), (32, 256, 501, 33), (32, 512, 501, 75)]:
    p = T.row_pitch(t)
    x = torch.randn(b, c, p, device="cuda").bfloat16()
    dy = torch.randn(b, c, p, device="cuda").bfloat16()
    y = torch.empty_like(x); dx = torch.empty_like(x)
    w = torch.randn(c, k, device="cuda")
    dw = torch.zeros(c, k, device="cuda")
    lens = torch.full((b,), t, dtype=torch.int32, device="cuda")
    calls = {
        "fwd": lambda: L.ts_train_dwconv_fwd(x.data_ptr(), lens.data_ptr(), lens.data_ptr(), w.data_ptr(), y.data_ptr(), b, c, t, t, k, 1, 1, (k - 1) // 2, p, p, 1, st),
        "bwd": lambda: L.ts_train_dwconv_bwd(dy.data_ptr(), x.data_ptr(), lens.data_ptr(), lens.data_ptr(), w.data_ptr(), dx.data_ptr(), dw.data_ptr(), b, c, t, t, k, 1, 1, (k - 1) // 2, p, p, 1, st),
    }
    for name, fn in calls.items():
        for _ in range(5):
            rc = fn()
        assert rc == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        gb = 2 * b * c * t * 2 * (1 if name == "fwd" else 1.5) / 1e9
        print(f"B={b} C={c} T={t} K={k} {name}: {us:7.1f} us  ({gb / us * 1e6 / 1e3:.2f} TB/s algorithmic)")
