"""Citrinet-1024 layer shapes (C3: 32 clips x 20 s) on the split kernel, next to the token-major GEMM of csrc/gemm_nt.hip and the vendor GEMM on the same
product: how far the fused launch is from a plain GEMM when the depthwise is 1-4 % of the work."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools.bench_tcs import layer, bench
from tools.diag.gemm_nt_check import run, frag

for T in (1001, 501, 251):
    for k in (1, 11, 21, 39):
        bench(f"T={T} 1024->1024 K{k}", layer(1024, 1024, k, 0, separable=k > 1), 32, T, iters=10)
    m = 32 * T
    x = torch.randn(m, 1024, device="cuda").to(torch.bfloat16)
    w = (torch.randn(1024, 1024, device="cuda") / 32).to(torch.bfloat16)
    bias = torch.randn(1024, device="cuda")
    frag(w)
    for name, fn in (("gemm_nt packed", lambda: run(x, w, bias, None, False, want32=False, packed=True)), ("vendor", lambda: torch.nn.functional.linear(x, w))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"T={T} token-major {m} x 1024 x 1024 {name:15s} {us:8.1f} us  {2 * m * 1024 * 1024 / us * 1e-6:7.1f} TFLOP/s")
