"""Merged kernel (csrc/tcs_v3.hip) vs the split kernel on the same tail-zero inputs, layer by layer (bit-level statistics), then timing.
    python tools/diag/v3_check.py [check|bench]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools.bench_tcs import layer, bench
from thunder_speech_amd import tensors as TS


def run(L, B, T, lens, v3, res=None):
    os.environ["TS_TCS_V3"] = "1" if v3 else "0"
    li = torch.tensor(lens, dtype=torch.int32, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    x = TS.backing(TS.pack(torch.randn(B, L.c_in, T, device="cuda", generator=g), li, slot="cx"))
    xr = TS.backing(TS.pack(torch.randn(B, L.c_res, T, device="cuda", generator=g), li, slot="cr")) if L.c_res else None
    out = TS.arena("co", B, L.c_out, T, "cuda")
    out.zero_()
    y, t_out = L.run(x, T, li, xr, T, li, out=out, in_tail_zero=True, zero_tail=True)
    torch.cuda.synchronize()
    return y[:, :, :T].float().clone()


def check():
    bad = 0
    cases = [(256, 256, 33, 0, 4, 751), (256, 256, 39, 256, 3, 400), (256, 512, 51, 0, 2, 751), (512, 512, 63, 0, 3, 751),
             (512, 512, 75, 512, 2, 600), (512, 512, 51, 256, 2, 751), (64, 64, 11, 0, 2, 100), (128, 320, 17, 64, 3, 193),
             (1024, 1024, 13, 0, 1, 500)]
    for (ci, co, k, res, B, T) in cases:
        L = layer(ci, co, k, res)
        lens = [T] + [max(1, T - 37 * (i + 1)) for i in range(B - 1)]
        a = run(L, B, T, lens, v3=False)
        b = run(L, B, T, lens, v3=True)
        d = (a - b).abs()
        scale = float(a.abs().max())
        nz = int((d > 0).sum())
        print(f"{ci}->{co} K{k} res {res} B{B} T{T}: max |split - v3| = {float(d.max()):.5f} (scale {scale:.2f}), "
              f"{nz} of {d.numel()} differ, tails zero: {bool((b[1, :, lens[1]:] == 0).all()) if B > 1 else True}")
        if float(d.max()) > 0.02 * max(scale, 1.0):
            bad += 1
            idx = torch.nonzero(d > 0.02 * max(scale, 1.0))[:5]
            print("   first bad:", idx.tolist())
    print("BAD" if bad else "OK")
    return bad


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    rc = 0
    if mode in ("check", "both"):
        rc = check()
    if mode in ("bench", "both"):
        for v3 in (False, True):
            os.environ["TS_TCS_V3"] = "1" if v3 else "0"
            print("merged kernel (v3)" if v3 else "split kernel")
            for (c_in, c, k, res) in [(256, 256, 33, 0), (256, 256, 39, 256), (256, 512, 51, 0), (512, 512, 51, 0), (512, 512, 63, 0),
                                      (512, 512, 63, 512), (512, 512, 75, 0)]:
                bench(f"{c_in}->{c} K{k} res {res}", layer(c_in, c, k, res), 64, 751)
    sys.exit(rc)
