"""Experiment: QuartzNet15x5 encoder (64 x 15 s) as ONE chain of full-batch launches vs N chains of 64/N-clip launches on N
streams (with / without a workgroup cap), all replayed from a hipGraph.  Encoder time per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import build_model
from thunder_speech_amd import plan, tensors as TS

dev = torch.device("cuda", 0)
module = build_model(dev)
B, S = 64, 15
wav = (0.1 * torch.randn(B, 16000 * S, generator=torch.Generator().manual_seed(1234))).to(dev)
lengths = torch.full((B,), 16000 * S, dtype=torch.int32, device=dev)
with torch.no_grad():
    feats, fl = module.audio_transform(wav, lengths)

def time_graph(fn, iters=50):
    with torch.no_grad():
        fn(); fn(); torch.cuda.synchronize()
        g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream(dev)
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                out = fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out

def single():
    return module.encoder(feats, fl)[0]

def make_split(n, cap):
    streams = [torch.cuda.Stream(dev) for _ in range(n)]
    per = B // n
    def run():
        cur = torch.cuda.current_stream(dev)
        outs = []
        plan.WG_LIMIT = cap
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                x = TS.tag_tail_zero(feats[i * per:(i + 1) * per])
                outs.append(module.encoder(x, fl[i * per:(i + 1) * per])[0])
        plan.WG_LIMIT = 0
        for s in streams: cur.wait_stream(s)
        return outs
    return run

t1, ref = time_graph(single)
print(f"1 chain, 256 workgroups: {t1:.3f} ms")
for n, cap in ((2, 0), (2, 128), (4, 64), (2, 160), (4, 0), (4, 128)):
    t, outs = time_graph(make_split(n, cap))
    ok = torch.equal(torch.cat(outs), ref)
    print(f"{n} chains on {n} streams, workgroup cap {cap or 256}: {t:.3f} ms  ({t1 / t:.3f}x)  identical output: {ok}")
