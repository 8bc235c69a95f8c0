import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools.diag.gemm_nt_check import run
for (m, n, k) in [(8192, 8192, 8192), (15984, 1024, 1024)]:
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16)
    for exp in [int(v) for v in os.environ.get('EXPS', '0,1,2,4,8,3,7,15').split(',')]:
        os.environ["TS_EXP"] = str(exp)
        for _ in range(3):
            run(x, w, None, None, False, want32=False, packed=os.environ.get('PACKED') == '1')
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(x, w, None, None, False, want32=False, packed=os.environ.get('PACKED') == '1')
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"m {m} n {n} k {k} exp={exp:2d}: {ms * 1e3:8.1f} us  {2 * m * n * k / ms * 1e-9:7.1f} TFLOP/s")
