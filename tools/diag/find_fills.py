"""Where do the at::native fill launches of a training step come from?  One eager QuartzNet15x5 step under torch.profiler with stacks;
prints the Python call sites of aten::fill_ / aten::zero_ / aten::zeros / aten::full, grouped."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from thunder_speech_amd import train_ops
from thunder_speech_amd.optim import FusedAdamW
from thunder_speech_amd.parallel import GradientSync
from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
from thunder_speech_amd.utils import variance_preserving_init_

dev = torch.device("cuda:0")
phase = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.manual_seed(0)
m = build_synthetic_quartznet(repeat_blocks=3)
variance_preserving_init_(m.encoder, m.decoder, seed=0)
m = m.to(dev).train()
if phase == 1:
    from thunder_speech_amd.callbacks import FinetuneEncoderDecoder
    FinetuneEncoderDecoder(train_batchnorm=True).freeze_before_training(m)
train_ops.set_activation_dtype("bf16")
g = torch.Generator().manual_seed(1)
wav = (0.1 * torch.randn(8, 160000, generator=g)).to(dev)
lengths = torch.full((8,), 160000.0, device=dev)
texts = ["hello world this is a test"] * 8
trainable = [p for p in m.parameters() if p.requires_grad]
opt, sync = FusedAdamW(trainable, lr=1e-3), GradientSync(trainable)


def step():
    sync.zero_grad()
    loss = m.training_step((wav, lengths, texts), 0)
    loss.backward()
    sync.finish()
    opt.step()


if len(sys.argv) > 2:                       # graphed: profile the call that captures (2 eager warm-up passes + the capture pass)
    from thunder_speech_amd.train_graph import GraphedTrainStep
    graphed = GraphedTrainStep(m, opt, sync, max_target_len=160)
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        graphed((wav, lengths, texts))
else:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        step()
torch.cuda.synchronize()
sites = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::full", "aten::zeros_like", "aten::ones", "aten::clone", "aten::copy_", "aten::add_", "aten::add", "aten::mul"):
        st = [s for s in (e.stack or []) if "thunder_speech_amd" in s or "torch/autograd" in s]
        sites[(e.name, tuple(st[:3]))] += 1
for (name, st), n in sites.most_common(40):
    print(n, name, " <- ".join(s.split("/root/repo/")[-1] for s in st))
