"""Encoder (C2, 64 x 15 s) hipGraph time of the tree this file is run from: python tools/enc_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda", 0)
m = bench.build_model(dev)
wav = (0.1 * torch.randn(64, 240000, generator=torch.Generator().manual_seed(1234))).to(dev)
ln = torch.full((64,), 240000, dtype=torch.int32, device=dev)
def time_graph(fn, steps):
    side = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
with torch.no_grad():
    f, fl = m.audio_transform(wav, ln)
    m.encoder(f, fl); torch.cuda.synchronize()
    print("encoder ms:", " ".join(f"{time_graph(lambda: m.encoder(f, fl), 40):.3f}" for _ in range(3)), "tree", ROOT, os.environ.get("TS_LIB_VARIANT", ""))
