"""A/B of the half-tile GEMM's 32- and 64-deep ring slots (experiment switch ts_exp_gemm_bk64), interleaved on one box."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from thunder_speech_amd import _lib
from tools.diag.gemm_nt_check import run, frag, check

L = _lib.lib()
sw = L.ts_exp_gemm_bk64
sw.argtypes = [ctypes.c_int]
sw.restype = None
rc = 0
for on in (0,):
    sw(on)
    print({0: "bk32", 1: "bk64", 2: "roll"}[on], flush=True)
    rc += check()
shapes = [(15984, 1024, 1024), (15984, 3072, 1024), (15984, 4096, 1024), (15984, 1024, 4096), (15984, 512, 1536), (8192, 8192, 8192)]
for (m, n, k) in shapes:
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda")
    frag(w)
    res = {}
    for rnd in range(3):
        for name in ("bk32", "noB", "noA", "noAB", "torch"):
            sw({"bk32": 0, "noB": 3, "noA": 4, "noAB": 5, "torch": 0}[name])
            fn = (lambda: torch.nn.functional.linear(x, w)) if name == "torch" else (lambda: run(x, w, bias, None, False, want32=False, packed=True))
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                fn()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 30 * 1e3)
    print(f"m {m} n {n} k {k}: " + "  ".join(f"{nm} {min(v):7.1f} us ({2 * m * n * k / min(v) * 1e-6:6.0f} TF)" for nm, v in res.items()), flush=True)
sw(1)
sys.exit(rc)
