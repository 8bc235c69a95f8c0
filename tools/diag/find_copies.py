"""List every GPU op of one eager inference step (torch.profiler), to find host-side copies between the fused launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from thunder_speech_amd.module import greedy_decode

dev = torch.device("cuda", 0)
m = bench.build_model(dev)
B, S = 64, 15
wav = (0.1 * torch.randn(B, 16000 * S)).to(dev)
lengths = torch.full((B,), 16000 * S, dtype=torch.int32, device=dev)
with torch.no_grad():
    for _ in range(2):
        greedy_decode(m(wav, lengths)[0])
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        greedy_decode(m(wav, lengths)[0])
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60))
for ev in prof.events():
    if "copy" in ev.name.lower() or "Memcpy" in ev.name:
        st = [s for s in (ev.stack or []) if "thunder_speech_amd" in s or "bench" in s][:3]
        print(ev.name, ev.device_type, st)
