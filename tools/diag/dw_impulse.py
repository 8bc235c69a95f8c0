"""Forensic: impulse input through an identity pointwise -- the output IS the tap vector the kernel used.  Reports, for
corrupted (clip, channel, tile)s, which 4-tap groups are wrong and which channel's taps they equal."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from thunder_speech_amd import plan, tensors as TS
C, K, T, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 2
torch.manual_seed(0)
dw = (torch.randn(C, 1, K) * 0.5).to(torch.bfloat16).float()
pw = torch.eye(C).reshape(C, C, 1)
bn = [torch.ones(C), torch.zeros(C), torch.zeros(C), torch.ones(C) - 1e-3]
layer = plan.make_tcs_layer("cuda", dw_w=dw, pw_w=pw, bn=bn, kernel=K, stride=1, dilation=1, padding=K // 2, relu=False)
x = torch.zeros(B, C, T); x[:, :, 48::96] = 1.0
li = torch.full((B,), T, dtype=torch.int32, device="cuda")
ref = torch.nn.functional.conv1d(x.cuda(), dw.cuda(), padding=K // 2, groups=C)
W = dw[:, 0, :].numpy()
dw2 = (torch.randn(C, 1, K) * 0.5).to(torch.bfloat16).float()
layer2 = plan.make_tcs_layer("cuda", dw_w=dw2, pw_w=pw, bn=bn, kernel=K, stride=1, dilation=1, padding=K // 2, relu=False)
W2 = dw2[:, 0, :].numpy()
ALT = bool(os.environ.get("ALT"))
shown = 0
for it in range(int(sys.argv[4])):
    xb = TS.backing(TS.pack(x.cuda(), li, slot=("d", 0)))
    if ALT:
        layer2.run(xb, T, li, out=TS.arena(("do2", 0), B, C, T, "cuda"), in_tail_zero=True, zero_tail=True)
    out = TS.arena(("do", 0), B, C, T, "cuda")
    y, _ = layer.run(xb, T, li, out=out, in_tail_zero=True, zero_tail=True)
    torch.cuda.synchronize()
    got = y[:, :, :T].float()
    err = (got - ref).abs()
    bad = (err > 0.01).nonzero()
    if len(bad) == 0:
        continue
    seen = set()
    for b, c, t in bad.tolist():
        key = (b, c, t // 96)
        if key in seen:
            continue
        seen.add(key)
    print(f"iter {it}: {len(seen)} corrupted (clip, channel, tile)s; chunks {sorted({c // 64 for _, c, _ in seen})}")
    for (b, c, tile) in sorted(seen)[:3]:
        ti = tile * 96 + 48
        used = np.array([float(got[b, c, ti + (K // 2) - k]) if 0 <= ti + (K // 2) - k < T else np.nan for k in range(K)])
        line = []
        for q0 in range(0, K, 4):
            seg = slice(q0, min(K, q0 + 4))
            if np.nanmax(np.abs(used[seg] - W[c, seg])) < 1e-3:
                line.append("ok")
                continue
            d = np.nanmax(np.abs(W[:, seg] - used[seg]), axis=1)
            d2 = np.nanmax(np.abs(W2[:, seg] - used[seg]), axis=1)
            src = int(np.argmin(d)); src2 = int(np.argmin(d2))
            if d2[src2] < d[src]:
                line.append(f"[{q0}:OTHER-LAYER ch{src2}(chunk{src2 // 64},+{src2 % 64}){'!' if d2[src2] > 1e-3 else ''}]")
            else:
                line.append(f"[{q0}:ch{src}(chunk{src // 64},+{src % 64}){'!' if d[src] > 1e-3 else ''}]")
        print(f"   clip {b} channel {c} (chunk {c // 64}, in-chunk {c % 64}) tile {tile}: " + " ".join(line))
    shown += 1
    if shown >= 4:
        break
