"""Which host-side ops of a fine-tuning step (phase 1) end up as device copies?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from collections import Counter
from thunder_speech_amd.optim import FusedAdamW
from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
from thunder_speech_amd.utils import variance_preserving_init_
m = build_synthetic_quartznet(repeat_blocks=3); variance_preserving_init_(m.encoder, m.decoder, seed=0); m = m.cuda().train()
m.encoder.eval()
for p in m.encoder.parameters(): p.requires_grad_(False)
opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3)
wav = (0.1 * torch.randn(32, 160000)).cuda(); lengths = torch.full((32,), 160000.0).cuda(); texts = ["hello world this is a test"] * 32
def step():
    opt.zero_grad(); loss = m.training_step((wav, lengths, texts), 0); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step(); torch.cuda.synchronize()
c = Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_"):
        st = [s for s in (ev.stack or []) if "thunder_speech_amd" in s or "tools/" in s]
        c[(ev.name, st[0] if st else (str(ev.input_shapes)[:60] if ev.input_shapes else "?"))] += 1
for k, v in c.most_common(25): print(v, k)
m = Counter()
for ev in prof.events():
    if 'Memcpy' in ev.name or 'memcpy' in ev.name.lower() or 'copyBuffer' in ev.name:
        m[ev.name] += 1
print(m)
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=12, max_name_column_width=50))
