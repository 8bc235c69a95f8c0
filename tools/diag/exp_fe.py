"""Timing experiments on the mel front end (diagnostic build -DTS_EXP): TS_EXP bits switch pieces of stft_mel_kernel off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from thunder_speech_amd.quartznet.transform import FilterbankFeatures

fb = FilterbankFeatures().cuda().eval()
wav = 0.1 * torch.randn(64, 240000, device="cuda")
lens = torch.full((64,), 240000.0, device="cuda")
for exp in [int(v) for v in os.environ.get("EXPS", "0,1,2,3,4,7").split(",")]:
    os.environ["TS_EXP"] = str(exp)
    for _ in range(3):
        fb(wav, lens)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fb(wav, lens)
    e1.record(); torch.cuda.synchronize()
    print(f"exp={exp}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per front end (both kernels + host launches)")
