"""Per-stage timestamps of workgroup 7 of the split kernel (diagnostic variant built with -DTS_STAMP by tools/variants.py):
    TS_LIB_VARIANT=stamp python tools/diag/stamp_dump.py 512 512 63 [res]
prints, for consumer waves 0 and 4 and producer wave 8, cycles per stage segment, and the in-kernel clock."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
dbg = torch.zeros(12 * 128, dtype=torch.int64, device="cuda")
os.environ["TS_DBG_PTR"] = str(dbg.data_ptr())
from tools.bench_tcs import layer
from thunder_speech_amd import tensors as TS
cin, cout, k = [int(v) for v in sys.argv[1:4]]
res = int(sys.argv[4]) if len(sys.argv) > 4 else 0
L = layer(cin, cout, k, res, separable=k > 1)
Bn, T = 64, 751
li = torch.full((Bn,), T, dtype=torch.int32, device="cuda")
x = TS.backing(TS.pack(torch.randn(Bn, cin, T, device="cuda"), li, slot="bx"))
xr = TS.backing(TS.pack(torch.randn(Bn, res, T, device="cuda"), li, slot="br")) if res else None
out = TS.arena("bo", Bn, cout, T, "cuda")
for _ in range(20):
    L.run(x, T, li, xr, T, li, out=out, in_tail_zero=True, zero_tail=True)
torch.cuda.synchronize()
d = dbg.cpu().view(12, 128)
for w in (0, 4, 8):
    r = d[w]
    dt_real, dt_clk = int(r[122] - r[120]), int(r[123] - r[121])
    print(f"wave {w}: clock ~ {dt_clk / max(dt_real, 1) * 100:.0f} MHz over {dt_clk} cycles")
print("consumer wave 0 / 4: [mfma-phase, barrier-wait] per stage;  producer wave 8: [wait, xs_write, begin+issue, passes, pack, barrier]")
c0, c4, p = d[0], d[4], d[8]
for s in range(14):
    a = [int(c0[8 * s + i]) for i in range(3)]
    b = [int(c4[8 * s + i]) for i in range(3)]
    q = [int(p[8 * s + i]) for i in range(7)]
    nxt = int(c0[8 * (s + 1)])
    print(f"  {s:2d}  c0: mfma {a[1]-a[0]:5d} bar {a[2]-a[1]:5d} gap {nxt-a[2]:5d} | c4: mfma {b[1]-b[0]:5d} bar {b[2]-b[1]:5d} | "
          f"p8: {q[1]-q[0]:5d} {q[2]-q[1]:5d} {q[3]-q[2]:5d} {q[4]-q[3]:5d} {q[5]-q[4]:5d} {q[6]-q[5]:5d}  stage {a[0]-int(c0[0]):7d}")
