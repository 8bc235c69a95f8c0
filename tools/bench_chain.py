"""A/B: QuartzNet15x5 encoder (C2: 64 x 15 s) with the repeats of a block as ONE chain launch vs one launch per sub-block, and the
same per block group.  python tools/bench_chain.py [--steps N]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from thunder_speech_amd import plan


def time_graph(fn, steps):
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    module = bench.build_model(dev)
    wav = (0.1 * torch.randn(args.batch, 16000 * 15, generator=torch.Generator().manual_seed(1234))).to(dev)
    lengths = torch.full((args.batch,), 16000 * 15, dtype=torch.int32, device=dev)
    with torch.no_grad():
        feats, fl = module.audio_transform(wav, lengths)
        outs = {}
        for chain in (False, True, "force", False, "force"):
            plan.CHAIN = chain
            module.encoder(feats, fl); torch.cuda.synchronize()
            ms = time_graph(lambda: module.encoder(feats, fl), args.steps)
            y, _ = module.encoder(feats, fl)
            torch.cuda.synchronize()
            outs[bool(chain)] = y.clone()
            print(f"encoder, chain={chain}: {ms:.3f} ms  (roofline frac {6.73e9 / (ms * 1e-3) / 8e12:.3f})", flush=True)
        print("bit-identical:", torch.equal(outs[True].view(torch.int16), outs[False].view(torch.int16)))
        # per block
        from thunder_speech_amd import tensors as TS
        x = feats
        blocks = list(module.encoder.children())
        with TS.lengths_scope():
            xi = TS.pack(feats, fl, slot=("bc", 0)) if not TS.is_internal(feats) else feats
            for i, blk in enumerate(blocks):
                res = []
                for chain in (False, "force"):
                    plan.CHAIN = chain
                    run = lambda: blk._run_fused(xi, fl, internal=True, slot=("bc", i % 2))
                    run(); torch.cuda.synchronize()
                    res.append(time_graph(run, args.steps))
                y, fl2, _ = blk._run_fused(xi, fl, internal=True, slot=("bc", i % 2))
                n = len(blk._cache.get(blk._params(), blk._compile))
                print(f"block {i:2d} ({n} launches): single {res[0] * 1e3:7.1f} us  chain {res[1] * 1e3:7.1f} us", flush=True)
                xi, fl = y, fl2
    plan.CHAIN = True


if __name__ == "__main__":
    main()
