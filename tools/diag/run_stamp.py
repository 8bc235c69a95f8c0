import sys, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from thunder_speech_amd import build as B
shutil.copy(os.path.join(ROOT, "tools", "diag", "libstamp.so"), B.lib_path())
dbg = torch.zeros(8 * 64, dtype=torch.int64, device="cuda")
os.environ["TS_DBG_PTR"] = str(dbg.data_ptr())
from tools.bench_tcs import layer
from thunder_speech_amd import _lib
cin, cout, k = [int(v) for v in sys.argv[1:4]]
L = layer(cin, cout, k, 0, separable=k > 1)
Bn, T = 64, 751
from thunder_speech_amd import tensors as TS
li = torch.full((Bn,), T, dtype=torch.int32, device="cuda")
x = TS.backing(TS.pack(torch.randn(Bn, cin, T, device="cuda"), li, slot="bx"))
out = TS.arena("bo", Bn, cout, T, "cuda")
for _ in range(3):
    L.run(x, T, li, out=out, in_tail_zero=True, zero_tail=True)
torch.cuda.synchronize()
d = dbg.cpu().view(8, 64)
n = (cin + 63) // 64
print("producer wave 4 (cycles since kernel-start stamp of consumer wave 0):")
base = int(d[0, 0])
p = d[4]
print("  start", int(p[0]) - base)
for s in range(n):
    a = [int(p[1 + s * 5 + i]) for i in range(5)]
    print(f"  stage {s}: begin@{a[0]-base:6d}  write_x/t(wait loads) {a[1]-a[0]:5d}  issue-next {a[2]-a[1]:5d}  dw {a[3]-a[2]:5d}  pack+barrier {a[4]-a[3]:5d}")
c = d[0]
print("consumer wave 0:")
for s in range(n):
    print(f"  stage {s}: wait-barrier {int(c[2+2*s])-int(c[1+2*s]):6d}  (barrier passed @{int(c[2+2*s])-base})   pw {int(c[1+2*(s+1)] if s+1<n else c[60])-int(c[2+2*s]):6d}")
print(f"  epilogue {int(c[61])-int(c[60])}, end@{int(c[61])-base}")

print("consumer wave 0 k-step stamps (cycles after barrier): ")
for s_ in range(min(n, 8)):
    t0_ = int(c[2 + 2 * s_])
    print("  stage", s_, [int(c[20 + 4 * s_ + i]) - t0_ for i in range(4)])
