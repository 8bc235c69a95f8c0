"""Which gradients of a QuartzNet15x5 training step differ between two runs from identical state (deterministic mode on)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from thunder_speech_amd import train_ops
from tools import train_margin_model as tmm

dev = torch.device("cuda", 0)
m = tmm.build_module(dev, 0).train()
train_ops.set_activation_dtype("bf16")
train_ops.set_deterministic(True, dev)
wav, lengths, texts = tmm.tone_clips(32, 10, 17, dev, kind="mix")
names = [n for n, p in m.named_parameters() if p.requires_grad]
params = [p for _, p in m.named_parameters() if p.requires_grad]
bufs0 = {k: v.clone() for k, v in m.state_dict().items()}

def run():
    m.load_state_dict(bufs0)
    torch.manual_seed(5)
    for p in params:
        p.grad = None
    loss = m.training_step((wav, lengths, texts), 0)
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().clone(), [p.grad.clone() for p in params], {k: v.clone() for k, v in m.state_dict().items() if "running" in k}

runs = [run() for _ in range(3)]
print("loss bits equal:", [bool(torch.equal(runs[0][0], r[0])) for r in runs[1:]], float(runs[0][0]))
bad = []
for i, n in enumerate(names):
    if not all(torch.equal(runs[0][1][i], r[1][i]) for r in runs[1:]):
        d = max(float((runs[0][1][i] - r[1][i]).abs().max()) for r in runs[1:])
        bad.append((n, tuple(params[i].shape), d))
print(len(bad), "of", len(names), "gradients differ")
for b in bad[:40]:
    print("  ", b)
rb = [k for k in runs[0][2] if not all(torch.equal(runs[0][2][k], r[2][k]) for r in runs[1:])]
print("running stats differing:", len(rb), rb[:5])
