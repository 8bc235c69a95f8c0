"""What the vendor GEMM (hipBLASLt through torch) needs for the pointwise products of the C2 layers, as plain GEMMs with nothing fused:
the practical ceiling of the matrix-core part of a fused TCS launch.  tokens = 64 clips x 751 frames."""
import torch
def t(m, n, k, iters=50):
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16); w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): torch.nn.functional.linear(a, w)
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.nn.functional.linear(a, w)
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters): torch.nn.functional.linear(a, w)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"M={m} N={n} K={k}: {us:7.1f} us  {2*m*n*k/us*1e-6:7.1f} TFLOP/s", flush=True)
for (n, k) in [(256, 256), (512, 256), (512, 512), (512, 1024), (1024, 512), (1024, 1024)]:
    t(64 * 751, n, k)
