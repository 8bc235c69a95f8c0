"""Random sweep of QuartznetBlock geometries that take the split TCS kernel (c_in % 64 == 0, tail-zero internal path) against the oracle's
bf16-ordered evaluation: channel counts, kernel sizes, repeat counts, clip counts, ragged lengths, frame counts around the tile sizes.
python tools/diag/split_sweep.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from oracle import tcs as otcs
from oracle.primitives import bf16_round
from thunder_speech_amd.quartznet.blocks import QuartznetBlock

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 0))
bad = 0
for case in range(n_cases):
    cin = int(rng.choice([64, 128, 256, 512]))
    cout = int(rng.choice([128, 256, 512, 640, 1024]))
    k = int(rng.choice([5, 11, 21, 33, 39, 51, 63, 75, 83]))
    repeat = int(rng.integers(1, 5))
    b = int(rng.integers(1, 7))
    t = int(rng.choice([95, 96, 97, 191, 192, 193, 300, 751, 1001, int(rng.integers(40, 1300))]))
    residual = bool(rng.integers(0, 2))
    lengths = torch.tensor([t] + [int(rng.integers(1, t + 1)) for _ in range(b - 1)])
    spec = otcs.BlockSpec(cin, cout, repeat=repeat, kernel=k, stride=1, dilation=1, residual=residual, separable=True)
    sd = otcs.synth_encoder_state([spec], seed=case)
    sd = {key[2:]: v for key, v in sd.items()}
    blk = QuartznetBlock(cin, cout, repeat=repeat, kernel_size=(k,), residual=residual, separable=True)
    blk.load_state_dict(sd, strict=True)
    blk = blk.cuda().eval()
    g = torch.Generator().manual_seed(1000 + case)
    x = bf16_round(torch.randn(b, cin, t, generator=g))
    want, want_len = otcs.block_forward(spec, sd, "", x, lengths, emulate_bf16=True)
    with torch.no_grad():
        got, got_len = blk(x.cuda(), lengths.cuda())
        got2, _ = blk(x.cuda(), lengths.cuda())          # the second call runs on warm arenas
    got = got.float().cpu()
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    same = torch.equal(got2.float().cpu(), got)
    ok = err <= 0.016 * scale and torch.equal(got_len.cpu(), want_len) and same
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} cin {cin} cout {cout} k {k} R {repeat} B {b} T {t} res {int(residual)}: max err {err:.4f} (scale {scale:.2f}) repeatable {same}")
print(f"{bad} of {n_cases} cases failed")
sys.exit(1 if bad else 0)
