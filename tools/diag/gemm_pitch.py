"""Does the output row pitch matter for the GEMM epilogue?  Same product, bf16 result, pitch N vs N + 64 elements."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from thunder_speech_amd import _lib
L = _lib.lib()
for (m, n, k) in [(8192, 8192, 1024), (15984, 4096, 1024), (15984, 1024, 1024), (15984, 3072, 1024)]:
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16)
    wf = torch.empty_like(w)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(L.ts_gemm_nt_pack_w(w.data_ptr(), k, n, k, wf.data_ptr(), s), "pack")
    for pad in (0, 64, 8):
        y16 = torch.empty(m, n + pad, dtype=torch.bfloat16, device="cuda")
        run = lambda: L.ts_gemm_nt_bf16_packed(x.data_ptr(), k, w.data_ptr(), k, wf.data_ptr(), None, None, 0, None, 0, y16.data_ptr(), n + pad, m, n, k, 0, s)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"m {m} n {n} k {k} out pitch n+{pad}: {ms * 1e3:7.1f} us  {2 * m * n * k / ms * 1e-9:7.1f} TFLOP/s")
