"""Diagnostic: depthwise through an identity pointwise, compared per (channel, frame) with torch (GPU) -- shows which
channel chunk / producer wave / time run a wrong tap or window belongs to."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from thunder_speech_amd import plan, tensors as TS
C, K, T, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 2
torch.manual_seed(0)
dw = torch.randn(C, 1, K) * 0.2
if os.environ.get("SAME_CHUNKS"):
    dw = dw[:64].repeat(C // 64, 1, 1)          # every 64-channel chunk has the same taps: stale-chunk reads become invisible
pw = torch.eye(C).reshape(C, C, 1)
bn = [torch.ones(C), torch.zeros(C), torch.zeros(C), torch.ones(C) - 1e-3]
layer = plan.make_tcs_layer("cuda", dw_w=dw, pw_w=pw, bn=bn, kernel=K, stride=1, dilation=1, padding=K // 2, relu=False)
x = torch.randn(B, C, T).to(torch.bfloat16).float()
li = torch.full((B,), T, dtype=torch.int32, device="cuda")
ref = torch.nn.functional.conv1d(x.cuda(), dw.to(torch.bfloat16).float().cuda(), padding=K // 2, groups=C)
bad_total = 0
for it in range(int(sys.argv[4]) if len(sys.argv) > 4 else 5):
    xb = TS.backing(TS.pack(x.cuda(), li, slot=("d", 0)))
    out = TS.arena(("do", 0), B, C, T, "cuda")
    y, _ = layer.run(xb, T, li, out=out, in_tail_zero=True, zero_tail=True)
    torch.cuda.synchronize()
    err = (y[:, :, :T].float() - ref).abs()
    bad = (err > 0.05).nonzero()
    bad_total += len(bad)
    if len(bad):
        b, c, t = bad[:, 0], bad[:, 1], bad[:, 2]
        print(f"iter {it}: {len(bad)} bad; clips {sorted(set(b.tolist()))} chunks {sorted(set((c // 64).tolist()))} "
              f"producers {sorted(set(((c % 64) // 16).tolist()))} channels-in-16 {sorted(set((c % 16).tolist()))[:16]} "
              f"tiles {sorted(set((t // 96).tolist()))} t%96 range {int((t % 96).min())}-{int((t % 96).max())} max err {float(err.max()):.3f}")
    else:
        print(f"iter {it}: ok (max err {float(err.max()):.4f})")
print("total bad", bad_total)

# ---- forensic: for one corrupted (clip, channel), recover the taps the kernel effectively used and say which chunk they came from
if os.environ.get("FORENSIC"):
    import numpy as np
    for it in range(200):
        xb = TS.backing(TS.pack(x.cuda(), li, slot=("d", 0)))
        out = TS.arena(("do", 0), B, C, T, "cuda")
        y, _ = layer.run(xb, T, li, out=out, in_tail_zero=True, zero_tail=True)
        torch.cuda.synchronize()
        err = (y[:, :, :T].float() - ref).abs()
        bad = (err > 0.05).nonzero()
        if len(bad) == 0:
            continue
        b, c, t = [int(v) for v in bad[0]]
        tile = t // 96
        ts = np.arange(tile * 96, min(T, tile * 96 + 96))
        xp = torch.nn.functional.pad(x[b, c], (K // 2, K // 2)).numpy().astype(np.float64)
        A = np.stack([xp[ts + k] for k in range(K)], axis=1)                   # [frames, K]
        got = y[b, c, ts].float().cpu().numpy().astype(np.float64)
        w_used, *_ = np.linalg.lstsq(A, got, rcond=None)
        wq = dw.to(torch.bfloat16).float().numpy()[:, 0, :]                    # [C, K]
        print(f"iter {it}: clip {b} channel {c} (chunk {c // 64}, in-chunk {c % 64}) tile {tile}")
        for q0 in range(0, K, 4):
            seg = slice(q0, min(K, q0 + 4))
            d = np.abs(wq[:, seg] - w_used[seg]).max(axis=1)
            best = int(np.argmin(d))
            print(f"   taps {q0:2d}-{min(K, q0 + 4) - 1:2d}: closest channel {best} (chunk {best // 64}, in-chunk {best % 64}) residual {d[best]:.3f}"
                  + ("" if best == c else "   <-- NOT this channel"))
        break
