"""ts_gemm_nt_bf16 (csrc/gemm_nt.hip) vs torch on the same bf16 operands, then timing against torch's own bf16 matmul (hipBLASLt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from thunder_speech_amd import _lib

L = _lib.lib()
dev = "cuda"


_FRAG = {}


def frag(w):
    key = (w.data_ptr(), tuple(w.shape))
    if key not in _FRAG:
        out = torch.empty_like(w)
        _lib.check(L.ts_gemm_nt_pack_w(w.data_ptr(), w.stride(0), w.shape[0], w.shape[1], out.data_ptr(), torch.cuda.current_stream().cuda_stream), "pack")
        _FRAG[key] = (out, w)
    return _FRAG[key][0]


def run(x, w, bias=None, res=None, gelu=False, want32=True, want16=True, packed=False):
    m, k = x.shape
    n = w.shape[0]
    y = torch.empty(m, n, dtype=torch.float32, device=dev) if want32 else None
    y16 = torch.empty(m, n, dtype=torch.bfloat16, device=dev) if want16 else None
    if packed:
        st = L.ts_gemm_nt_bf16_packed(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), frag(w).data_ptr(), bias.data_ptr() if bias is not None else None,
                                      res.data_ptr() if res is not None else None, res.stride(0) if res is not None else 0,
                                      y.data_ptr() if y is not None else None, n, y16.data_ptr() if y16 is not None else None, n, m, n, k, int(gelu),
                                      torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "ts_gemm_nt_bf16_packed")
        return y, y16
    st = L.ts_gemm_nt_bf16(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), bias.data_ptr() if bias is not None else None,
                           res.data_ptr() if res is not None else None, res.stride(0) if res is not None else 0,
                           y.data_ptr() if y is not None else None, n, y16.data_ptr() if y16 is not None else None, n, m, n, k, int(gelu),
                           torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "ts_gemm_nt_bf16")
    return y, y16


def check():
    bad = 0
    g = torch.Generator(device=dev).manual_seed(0)
    for (m, n, k, gelu, use_res) in [(256, 256, 64, False, False), (300, 96, 32, False, True), (1000, 1024, 1024, True, False),
                                     (15984, 1024, 4096, False, True), (999, 512, 1536, True, False), (257, 4096, 1024, True, False),
                                     (15984, 3072, 1024, True, False), (9000, 2048, 256, False, True), (8200, 2080, 128, False, False)]:
        x = torch.randn(m, k, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev, generator=g) / k ** 0.5).to(torch.bfloat16)
        bias = torch.randn(n, device=dev, generator=g)
        res = torch.randn(m, n, device=dev, generator=g) if use_res else None
        y, y16 = run(x, w, bias, res, gelu)
        yp, yp16 = run(x, w, bias, res, gelu, packed=True)
        same = torch.equal(y, yp) and torch.equal(y16, yp16)
        bad += not same
        ref = x.double() @ w.double().t() + bias.double()
        if gelu:
            ref = torch.nn.functional.gelu(ref)
        if use_res:
            ref = ref + res.double()
        err = float((y.double() - ref).abs().max())
        err16 = float((y16.double() - ref).abs().max())
        scale = float(ref.abs().max())
        ok = err <= 2e-4 * max(scale, 1) and err16 <= 1e-2 * max(scale, 1)
        bad += not ok
        print(f"m {m} n {n} k {k} gelu {gelu} res {use_res}: max err f32 {err:.2e}, bf16 {err16:.2e} (scale {scale:.2f}) {'ok' if ok else 'BAD'}, packed {'identical' if same else 'DIFFERS'}")
    return bad


def bench():
    for (m, n, k) in [(15984, 1024, 1024), (15984, 3072, 1024), (15984, 4096, 1024), (15984, 1024, 4096), (15984, 512, 1536), (8192, 8192, 8192)]:
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
        bias = torch.randn(n, device=dev)
        frag(w)
        for name, fn in (("ours", lambda: run(x, w, bias, None, False, want32=False)), ("packed", lambda: run(x, w, bias, None, False, want32=False, packed=True)),
                         ("torch", lambda: torch.nn.functional.linear(x, w))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print(f"{name:6s} m {m} n {n} k {k}: {ms * 1e3:8.1f} us  {2 * m * n * k / ms * 1e-9:7.1f} TFLOP/s")


if __name__ == "__main__":
    rc = check()
    bench()
    sys.exit(rc)
