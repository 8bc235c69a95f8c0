"""Race screen for the GEMM's operand ring (csrc/gemm_nt.hip): a DMA that lands late or a slot refilled early shows up as a wrong tile only now
and then, so every shape is run many times -- both operand paths, with a bandwidth-hungry copy running on a second stream to perturb the
timing -- and every result is compared bit for bit with the first one and with the other path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools.diag.gemm_nt_check import run, frag

torch.manual_seed(0)
noise_a = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
noise_b = torch.empty_like(noise_a)
side = torch.cuda.Stream()
bad = 0
iters = int(os.environ.get("ITERS", "150"))
for (m, n, k) in [(15984, 1024, 1024), (15984, 4096, 1024), (15984, 1024, 4096), (4000, 512, 1536), (999, 1024, 96), (8192, 8192, 512), (300, 3072, 64)]:
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda")
    frag(w)
    ref = run(x, w, bias, None, True, want32=False)[1].clone()
    wrong = 0
    for it in range(iters):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                noise_b.copy_(noise_a, non_blocking=True)
        for packed in (False, True):
            y = run(x, w, bias, None, True, want32=False, packed=packed)[1]
            wrong += int(not torch.equal(y, ref))
    torch.cuda.synchronize()
    bad += wrong
    print(f"m {m} n {n} k {k}: {2 * iters} runs, {wrong} differ from the first result")
sys.exit(1 if bad else 0)
