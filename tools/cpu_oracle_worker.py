#!/usr/bin/env python3
"""One process of bench.py's `cpu_baseline` leg: the CPU oracle (oracle/, a port of the reference path: fp32 torch-CPU ops) on this
process's own clips of the headline workload, `--threads` torch threads.  bench.py starts P of these BEFORE it touches the GPU (children of
a process that has initialised HIP are avoided altogether), over disjoint clips of the batch, and sums their throughput.

Protocol (stdin / stdout, line based): the worker builds the model (same seeds as bench.build_model: the same weights), draws its own
clips (0.1 * randn, as bench.py's batch), runs ONE warm-up pass, prints "READY"; on "GO" it runs --iters timed passes and
prints one JSON line {"seconds": wall time of the timed passes, "clips": n, "iters": k}.  This file is test / measurement infrastructure
(the oracle is the checker and the reported CPU baseline, never the product path).

    python tools/cpu_oracle_worker.py --index 0 --clips 4 --first 0 --threads 16 --iters 3 --seconds 15
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=0, help="index of this worker's first clip in the batch")
    ap.add_argument("--clips", type=int, default=4)
    ap.add_argument("--seconds", type=int, default=15)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--repeat-blocks", type=int, default=3)
    args = ap.parse_args()
    os.environ.setdefault("OMP_NUM_THREADS", str(args.threads))
    import torch
    torch.set_num_threads(args.threads)
    from oracle import decode as odec, frontend as ofe, tcs as otcs
    from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
    from thunder_speech_amd.utils import variance_preserving_init_
    module = build_synthetic_quartznet(repeat_blocks=args.repeat_blocks)
    variance_preserving_init_(module.encoder, module.decoder, seed=0)
    arch = otcs.quartznet_arch(repeat_blocks=args.repeat_blocks)
    sd = {k: v.detach() for k, v in module.encoder.state_dict().items()}
    dsd = {k: v.detach() for k, v in module.decoder.state_dict().items()}
    # this worker's own clips of the workload (same distribution as bench.py's batch, disjoint from every other worker's: seeded by `first`)
    g = torch.Generator().manual_seed(args.seed + 7919 * (args.first + 1))
    n = 16000 * args.seconds
    wav = 0.1 * torch.randn(args.clips, n, generator=g)
    lengths = torch.full((args.clips,), n)

    def run():
        with torch.no_grad():
            feats, fl = ofe.filterbank_features(wav, lengths)
            enc, _ = otcs.encoder_forward(arch, sd, feats, fl)
            logits = otcs.conv1d_decoder_forward(dsd, enc)
            ids = logits.argmax(1).numpy()
            return [odec.collapse_repeats(r) for r in ids]

    run()
    print("READY", flush=True)
    if sys.stdin.readline().strip() != "GO":
        return 2
    t0 = time.perf_counter()
    for _ in range(args.iters):
        run()
    dt = time.perf_counter() - t0
    print(json.dumps({"seconds": dt, "clips": args.clips, "iters": args.iters, "threads": torch.get_num_threads()}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
