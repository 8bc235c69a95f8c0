"""Time the fused TCS kernel on the QuartzNet15x5 C2 layer shapes (B=64, T'=751)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from thunder_speech_amd import _lib, plan

def layer(cin, cout, k, res, stride=1, dil=1, separable=True):
    g = torch.Generator().manual_seed(0)
    pad = (dil * (k - 1) + 1) // 2 if dil > 1 else k // 2
    bn = [torch.ones(cout), torch.zeros(cout), torch.zeros(cout), torch.ones(cout)]
    kw = dict(dw_w=torch.randn(cin, 1, k, generator=g) * 0.1 if separable else None,
              pw_w=torch.randn(cout, cin, 1, generator=g) * 0.05, bn=bn, kernel=k, stride=stride, dilation=dil,
              padding=pad, relu=True)
    if res:
        kw.update(res_w=torch.randn(cout, res, 1, generator=g) * 0.05, res_bn=bn, res_stride=1)
    return plan.make_tcs_layer("cuda", **kw)

def bench(name, L, B, T, iters=20, tz=True):
    from thunder_speech_amd import tensors as TS
    li = torch.full((B,), T, dtype=torch.int32, device="cuda")
    t_out = L.out_size(T)
    lr = torch.full((B,), t_out, dtype=torch.int32, device="cuda")
    x = TS.backing(TS.pack(torch.randn(B, L.c_in, T, device="cuda"), li, slot="bx"))          # tail-zero arena buffers
    xr = TS.backing(TS.pack(torch.randn(B, L.c_res, t_out, device="cuda"), lr, slot="br")) if L.c_res else None
    out = TS.arena("bo", B, L.c_out, t_out, "cuda")
    run = lambda: L.run(x, T, li, xr, t_out, lr, out=out, in_tail_zero=tz, zero_tail=tz)
    for _ in range(3): run()
    torch.cuda.synchronize()
    # capture the launches in a graph so that host (Python/ctypes) launch overhead is not measured
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(gr, stream=side):
            for _ in range(iters): run()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    macs = B * t_out * (L.c_in * L.c_out + (L.c_in * L.kernel if L.depthwise else 0) + L.c_res * L.c_out)
    byts = 2 * B * (T * L.c_in + t_out * L.c_out + t_out * L.c_res)
    print(f"{name:34s} {ms*1e3:8.1f} us  {2*macs/ms*1e-9:7.1f} TFLOP/s  {byts/ms*1e-6:7.1f} GB/s (algorithmic)")
    return ms

if __name__ == "__main__":
    B, T = 64, 751
    total = 0.0
    total += bench("stem 64->256 K33 s2", layer(64, 256, 33, 0, stride=2), B, 1501)
    for (c_in, c, k) in [(256, 256, 33), (256, 256, 39), (256, 512, 51), (512, 512, 63), (512, 512, 75)]:
        first = bench(f"{c_in}->{c} K{k}", layer(c_in, c, k, 0), B, T)
        mid = bench(f"{c}->{c} K{k}", layer(c, c, k, 0), B, T) if c_in != c else first
        last = bench(f"{c}->{c} K{k} + res({c_in})", layer(c, c, k, c_in), B, T)
        # block = first + 3 mid + last ; 3 blocks per group, blocks 2,3 have c_in == c
        lastcc = bench(f"{c}->{c} K{k} + res({c})", layer(c, c, k, c), B, T) if c_in != c else last
        total += (first + 3 * mid + last) + 2 * (4 * mid + lastcc)
    total += bench("512->512 K87 d2", layer(512, 512, 87, 0, dil=2), B, T)
    total += bench("512->1024 K1", layer(512, 1024, 1, 0, separable=False), B, T)
    print(f"estimated QuartzNet15x5 encoder time for 64x15 s: {total:.3f} ms -> {64*15/total*1e3:,.0f} audio-s/s; "
          f"HBM-roofline fraction (6.73 GB / 8 TB/s = 0.841 ms): {0.841/total:.3f}")
