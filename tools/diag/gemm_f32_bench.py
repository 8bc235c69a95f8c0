"""ts_gemm_f32 (csrc/gemm_f32.hip) vs the vendor library (torch f32 matmul = rocBLAS / hipBLASLt) on the shapes of the f32 modes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from thunder_speech_amd import _lib
torch.backends.cuda.matmul.allow_tf32 = False
L = _lib.lib()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
s = torch.cuda.current_stream().cuda_stream
for (m, n, k) in [(15984, 1024, 1024), (15984, 4096, 1024), (15984, 1024, 4096), (16032, 512, 512), (4096, 4096, 4096)]:
    a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda"); c = torch.empty(m, n, device="cuda")
    own = timeit(lambda: L.ts_gemm_f32(a.data_ptr(), k, 1, 0, 0, w.data_ptr(), 1, k, 0, 0, c.data_ptr(), n, 0, None, m, n, k, 1, 1, 0, 0, 0, s))
    ref = timeit(lambda: torch.nn.functional.linear(a, w))
    print(f"NT {m}x{n}x{k}: own {own:8.1f} us ({2*m*n*k/own*1e-6:6.1f} TF)   vendor {ref:8.1f} us ({2*m*n*k/ref*1e-6:6.1f} TF)", flush=True)
# per-clip conv products of the training path: v[b] = W u[b], 32 clips x 501 frames
for c in (256, 512):
    b, t, p = 32, 501, 512
    w = torch.randn(c, c, device="cuda"); u = torch.randn(b, c, p, device="cuda"); v = torch.empty(b, c, p, device="cuda")
    own = timeit(lambda: L.ts_gemm_f32(w.data_ptr(), c, 1, 0, 0, u.data_ptr(), p, 1, c * p, 0, v.data_ptr(), p, c * p, None, c, t, c, 1, b, 0, 0, 0, s))
    ref = timeit(lambda: torch.matmul(w, u))
    print(f"conv fwd {c}x{c} x [32 x {t}]: own {own:7.1f} us  vendor {ref:7.1f} us", flush=True)
