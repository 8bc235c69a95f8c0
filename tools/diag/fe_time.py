#!/usr/bin/env python3
"""Front end (FilterbankFeatures, 64 x 15 s) hipGraph time of the library variant named by TS_LIB_VARIANT ('' = product)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda", 0)
m = bench.build_model(dev)
wav = (0.1 * torch.randn(64, 240000, generator=torch.Generator().manual_seed(1234))).to(dev)
ln = torch.full((64,), 240000, dtype=torch.int32, device=dev)
with torch.no_grad():
    m.audio_transform(wav, ln); torch.cuda.synchronize()
    side = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.audio_transform(wav, ln)
        with torch.cuda.graph(g, stream=side):
            m.audio_transform(wav, ln)
    g.replay(); torch.cuda.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): g.replay()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 200 * 1e3)
print(f"front end us ({os.environ.get('TS_LIB_VARIANT', '') or 'product'}): " + " ".join(f"{v:.1f}" for v in res))
