import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import frontend as ofe, tcs as otcs
arch = otcs.quartznet_arch(repeat_blocks=3)
sd = otcs.synth_encoder_state(arch, seed=0)
wav = 0.1 * torch.randn(4, 240000); lengths = torch.full((4,), 240000)
print("cpu_count", os.cpu_count())
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    with torch.no_grad():
        feats, fl = ofe.filterbank_features(wav, lengths)
        t0 = time.perf_counter(); otcs.encoder_forward(arch, sd, feats, fl); dt = time.perf_counter() - t0
        t0 = time.perf_counter(); otcs.encoder_forward(arch, sd, feats, fl); dt = time.perf_counter() - t0
    print(nt, "threads:", round(60 / dt, 1), "audio-s/s", flush=True)
