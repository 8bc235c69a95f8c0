"""Why there is no margin-calibrated (strict all-frame argmax / transcript equality) fixture for QuartzNet15x5 at the C2 size (VERDICT round 3,
item 3), in numbers.  CPU only (the fp32 oracle and its bf16-ordered evaluation on one 15 s clip).

A fitted decoder can give every frame a fat top-1 / top-2 margin only if the 751 frames of a clip are linearly distinguishable in the
1024-channel encoder output BY MORE THAN the bf16 path's deviation from fp32.  Per singular direction of the (centred) fp32 encoder output of
a clip: signal = the direction's rms over the frames, noise = rms of (bf16-ordered oracle - fp32 oracle) projected on it.

  * calibrated synthetic weights (oracle.tcs.synth_encoder_state(calibrate=True), the test fixtures' weights): the 18-block random stack
    amplifies perturbations ~1.5x per block (oracle vs the REAL reference, both fp32: 6.6e-5 at the features -> 5.9e-2 at the output), so the
    bf16 evaluation decorrelates completely: noise rms = signal rms, no direction with SNR > 5;
  * bench.py's variance-preserving weights: stable (noise 0.35 % of the signal) but contractive: ~20 directions with SNR > 20, ~90 with
    SNR > 5 -- 751 frames cannot be told apart, a ridge decoder misclassifies half of them in fp32 already;
  * blends of the two and the `gain` knob stay on one side or the other.

QuartzNet5x5 (6 blocks) sits in between, which is why its fixture (tests/golden/qn5x5_e2e_margin.npz) exists.  With pretrained weights
(trained networks are neither chaotic nor rank-deficient) the reference's own golden transcript would be the test; they need the network.

    python tools/diag/margin_study.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import frontend as ofe, tcs as otcs          # noqa: E402
from oracle.primitives import bf16_round                 # noqa: E402


def study(name, arch, sd, feats, fl):
    with torch.no_grad():
        e32, _ = otcs.encoder_forward(arch, sd, feats, fl)
        e16, _ = otcs.encoder_forward(arch, sd, bf16_round(feats), fl, emulate_bf16=True)
    x = e32[0].t().double()
    xc = x - x.mean(0, keepdim=True)
    d = (e16 - e32)[0].t().double()
    _, sv, vh = torch.linalg.svd(xc, full_matrices=False)
    noise = (d @ vh.t()).pow(2).mean(0).sqrt()
    snr = sv / np.sqrt(x.shape[0]) / noise
    print(f"{name:34s} output rms {float(x.pow(2).mean().sqrt()):.3f}  bf16-vs-fp32 deviation rms {float(d.pow(2).mean().sqrt()):.4f}  "
          f"directions with SNR > 50 / 20 / 5: {int((snr > 50).sum())} / {int((snr > 20).sum())} / {int((snr > 5).sum())}  of {x.shape[0]} frames",
          flush=True)


def main():
    from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
    from thunder_speech_amd.utils import variance_preserving_init_
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    rng = np.random.Generator(np.random.PCG64(20260401))
    wav = torch.from_numpy((0.1 * rng.standard_normal((1, 240000))).astype(np.float32))
    wl = torch.tensor([240000.0])
    feats, fl = ofe.filterbank_features(wav, wl)
    for rb, tag in ((3, "15x5"), (1, "5x5")):
        arch = otcs.quartznet_arch(repeat_blocks=rb)
        m = build_synthetic_quartznet(repeat_blocks=rb)
        variance_preserving_init_(m.encoder, m.decoder, seed=0)
        study(f"QN{tag} variance-preserving", arch, {k: v.detach().clone().float() for k, v in m.encoder.state_dict().items()}, feats, fl)
        study(f"QN{tag} calibrated", arch, otcs.synth_encoder_state(arch, seed=0, calibrate=True), feats, fl)


if __name__ == "__main__":
    main()
