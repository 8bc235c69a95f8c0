"""Race screen for the chunked CTC recursion (csrc/ctc.hip): the same problem many times, with a copy running on a second stream; loss and
gradient must come out bit-identical every time (the kernel has no atomics on its serial path: any difference is a synchronisation bug)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from thunder_speech_amd.ctc_loss import calculate_ctc

g = torch.Generator().manual_seed(0)
noise_a = torch.empty(128 << 20, dtype=torch.uint8, device="cuda")
noise_b = torch.empty_like(noise_a)
side = torch.cuda.Stream()
bad = 0
for (B, V, T, smin, smax, SM) in [(32, 29, 501, 60, 140, 160), (8, 1024, 251, 20, 120, 128), (4, 12, 1500, 500, 700, 700), (16, 29, 37, 1, 18, 20)]:
    logits = torch.randn(B, V, T, generator=g).cuda()
    tl = torch.randint(smin, smax + 1, (B,), generator=g)
    y = torch.zeros(B, SM, dtype=torch.int64)
    for b in range(B):
        y[b, : tl[b]] = torch.randint(0, V - 1, (int(tl[b]),), generator=g)
    il = torch.randint(max(T // 2, 1), T + 1, (B,), generator=g)
    y, tl, il = y.cuda(), tl.cuda(), il.cuda()
    ref = None
    wrong = 0
    for it in range(int(os.environ.get("ITERS", "200"))):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                noise_b.copy_(noise_a, non_blocking=True)
        x = logits.clone().requires_grad_(True)
        loss = calculate_ctc(x, y, il, tl, V - 1)
        loss.backward()
        cur = (loss.detach().clone(), x.grad.clone())
        if ref is None:
            ref = cur
        wrong += int(not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])))
    torch.cuda.synchronize()
    bad += wrong
    print(f"B {B} V {V} T {T} S<={smax}: {wrong} runs differ from the first")
sys.exit(1 if bad else 0)
