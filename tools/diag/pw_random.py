"""Diagnostic: delta depthwise (dw(x) = x) + random pointwise; where do corrupted outputs sit?  All output channels of a
tile -> a producer wrote wrong rows; 64 output channels -> one consumer wave used a wrong A or W fragment."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from thunder_speech_amd import plan, tensors as TS
C, K, T, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 2
torch.manual_seed(0)
dw = torch.zeros(C, 1, K); dw[:, 0, K // 2] = 1.0
pw = (torch.randn(C, C, 1) / C ** 0.5).to(torch.bfloat16).float()
bn = [torch.ones(C), torch.zeros(C), torch.zeros(C), torch.ones(C) - 1e-3]
layer = plan.make_tcs_layer("cuda", dw_w=dw, pw_w=pw, bn=bn, kernel=K, stride=1, dilation=1, padding=K // 2, relu=False)
x = torch.randn(B, C, T).to(torch.bfloat16).float()
li = torch.full((B,), T, dtype=torch.int32, device="cuda")
ref = torch.einsum("oc,bct->bot", pw[:, :, 0].cuda(), x.cuda())
tot = 0
for it in range(int(sys.argv[4]) if len(sys.argv) > 4 else 5):
    xb = TS.backing(TS.pack(x.cuda(), li, slot=("d", 0)))
    out = TS.arena(("do", 0), B, C, T, "cuda")
    y, _ = layer.run(xb, T, li, out=out, in_tail_zero=True, zero_tail=True)
    torch.cuda.synchronize()
    err = (y[:, :, :T].float() - ref).abs()
    bad = (err > 0.1).nonzero()
    tot += len(bad)
    if len(bad):
        b, c, t = bad[:, 0], bad[:, 1], bad[:, 2]
        print(f"iter {it}: {len(bad)} bad; clips {sorted(set(b.tolist()))} co/64 {sorted(set((c // 64).tolist()))} n_co {len(set(c.tolist()))} "
              f"tiles {sorted(set((t // 96).tolist()))} max err {float(err.max()):.3f}")
print("total bad", tot)
