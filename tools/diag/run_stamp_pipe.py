"""Diagnostic: per-stage s_memtime stamps of workgroup 7 of the pipelined TCS kernel (needs tools/diag/libstamp.so)."""
import sys, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from thunder_speech_amd import build as B
shutil.copy(os.path.join(ROOT, "tools", "diag", os.environ.get("TS_STAMP_LIB", "libstamp.so")), B.lib_path())
dbg = torch.zeros(8 * 128, dtype=torch.int64, device="cuda")
os.environ["TS_DBG_PTR"] = str(dbg.data_ptr())
from tools.bench_tcs import layer
from thunder_speech_amd import _lib, tensors as TS
cin, cout, k = [int(v) for v in sys.argv[1:4]]
L = layer(cin, cout, k, 0)
Bn, T = 64, 751
li = torch.full((Bn,), T, dtype=torch.int32, device="cuda")
x = TS.backing(TS.pack(torch.randn(Bn, cin, T, device="cuda"), li, slot="bx"))
out = TS.arena("bo", Bn, cout, T, "cuda")
for _ in range(3):
    L.run(x, T, li, out=out, in_tail_zero=True, zero_tail=True)
torch.cuda.synchronize()
d = dbg.cpu().view(8, 128)
n = cin // 64
for w in (0, 5):
    p = d[w]; base = int(p[0])
    print(f"wave {w}:")
    for g in range(min(2 * n, 16)):
        a = [int(p[4 * g + i]) for i in range(4)]
        nxt = int(p[4 * g + 4]) if 4 * g + 4 < 64 else 0
        line = f"  stage {g:2d}: begin@{a[0]-base:7d} body {a[1]-a[0]:5d}  barrier {a[2]-a[1]:5d}"
        if a[3]:
            line += f"  | epilogue+init {nxt - a[3]:5d}" if nxt else "  | epilogue"
        print(line)
for w in (0, 5):
    p = d[w]
    e0 = int(p[4 * (n - 1) + 3])
    print(f"wave {w} epilogue of tile 0: start->len/partial {int(p[100])-e0}, nt0 pack {int(p[101])-int(p[100])}, nt0 store {int(p[102])-int(p[101])}, "
          f"nt1 pack {int(p[103])-int(p[102])}, nt1 store {int(p[104])-int(p[103])}, ->next stage begin {int(p[4*n])-int(p[104])}")
for w in (0, 5):
    p = d[w]
    t = [int(p[110 + i]) for i in range(8)]
    print(f"wave {w} stage 2 pass starts (rel. body begin {int(p[8])}):", [x - int(p[8]) for x in t if x], "body end", int(p[9]) - int(p[8]))
