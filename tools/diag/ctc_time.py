"""Time ts_ctc_loss on the C4 shape (32 utterances, 501 frames, 29 classes, 60-140 labels, s_max 160) and check it against F.ctc_loss."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from thunder_speech_amd.ctc_loss import calculate_ctc

dev = "cuda"
g = torch.Generator().manual_seed(0)
B, V, T, SM = 32, 29, 501, 160
logits = torch.randn(B, V, T, generator=g).to(dev).requires_grad_(True)
tl = torch.randint(60, 141, (B,), generator=g)
y = torch.zeros(B, SM, dtype=torch.int64)
for b in range(B):
    y[b, : tl[b]] = torch.randint(0, V - 1, (int(tl[b]),), generator=g)
il = torch.full((B,), T, dtype=torch.int32)
il[3] = 400; il[7] = 141
y, tl, il = y.to(dev), tl.to(dev), il.to(dev)
loss = calculate_ctc(logits, y, il, tl, V - 1)
loss.backward()
ref_in = logits.detach().cpu().double().requires_grad_(True)
ref = torch.nn.functional.ctc_loss(ref_in.permute(2, 0, 1).log_softmax(2), y.cpu(), il.cpu().long(), tl.cpu().long(), blank=V - 1, reduction="mean", zero_infinity=True)
ref.backward()
print("loss", float(loss), float(ref), "grad max err", float((logits.grad.cpu().double() - ref_in.grad).abs().max()), "grad scale", float(ref_in.grad.abs().max()))
x = logits.detach()
for _ in range(3):
    calculate_ctc(x, y, il, tl, V - 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
xg = x.clone().requires_grad_(True)
def run():
    l = calculate_ctc(xg, y, il, tl, V - 1)
    return l
run(); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph(); side = torch.cuda.Stream()
with torch.cuda.stream(side):
    run()
    with torch.cuda.graph(gr, stream=side):
        for _ in range(10): run()
gr.replay(); torch.cuda.synchronize()
e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
print(f"ts_ctc_loss (loss + gradient kernels): {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per call")
