"""Per-phase timestamps of one workgroup of the split kernel (diagnostic build: TS_CXXFLAGS=-DTS_STAMP).
python tools/diag/run_stamp_split.py 512 512 63"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TS_CXXFLAGS"] = "-DTS_STAMP"
from thunder_speech_amd import build
build.build(force=True, verbose=False)
import torch
dbg = torch.zeros(12 * 128, dtype=torch.int64, device="cuda")
os.environ["TS_DBG_PTR"] = str(dbg.data_ptr())
from tools.bench_tcs import layer
from thunder_speech_amd import tensors as TS
cin, cout, k = [int(v) for v in sys.argv[1:4]]
L = layer(cin, cout, k, 0, separable=k > 1)
Bn, T = 64, 751
li = torch.full((Bn,), T, dtype=torch.int32, device="cuda")
x = TS.backing(TS.pack(torch.randn(Bn, cin, T, device="cuda"), li, slot="bx"))
out = TS.arena("bo", Bn, cout, T, "cuda")
for _ in range(3):
    L.run(x, T, li, out=out, in_tail_zero=True, zero_tail=True)
torch.cuda.synchronize()
d = dbg.cpu().view(12, 128)
n = min((cin + 63) // 64 * 2, 15)
p, c = d[8], d[0]
base = int(p[0])
print("ticks are s_memtime units (100 MHz domain on this part? compare with the 8-stage total); producer wave 8:")
print(" stage   top  wait-loads  xs_write  begin+issue   passes  pack+write  barrier-wait   | consumer wave 0: mfma-issue  barrier-wait")
for s in range(n):
    a = [int(p[8 * s + i]) for i in range(7)]
    cc = [int(c[8 * s + i]) for i in range(3)]
    print(f"  {s:3d} {a[0]-base:7d} {a[1]-a[0]:9d} {a[2]-a[1]:9d} {a[3]-a[2]:11d} {a[4]-a[3]:9d} {a[5]-a[4]:10d} {a[6]-a[5]:12d}   | {cc[1]-cc[0]:24d} {cc[2]-cc[1]:12d}")
os.environ.pop("TS_CXXFLAGS", None)
build.build(force=True, verbose=False)            # put the product build back
