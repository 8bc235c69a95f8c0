"""How do rocBLAS GEMMs of the training pointwise convs behave with the odd leading dimension T' = 501 (fp32 vs bf16, 501 vs 504)?"""
import torch, time
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for t in (501, 504, 512):
    for dt in (torch.float32, torch.bfloat16):
        u = torch.randn(32, 512, t, device="cuda", dtype=dt)
        w = torch.randn(512, 512, device="cuda", dtype=dt)
        dv = torch.randn(32, 512, t, device="cuda", dtype=dt)
        f = bench(lambda: torch.matmul(w, u))                      # forward
        bd = bench(lambda: torch.matmul(w.t(), dv))                # backward data
        bw = bench(lambda: torch.bmm(dv, u.transpose(1, 2)))       # backward weight (per clip)
        fl = 2 * 32 * 512 * 512 * t
        print(f"T={t} {str(dt):15s} fwd {f:7.1f} us ({fl/f/1e6:6.1f} TFLOP/s)  bwd-data {bd:7.1f} us  bwd-weight {bw:7.1f} us")
