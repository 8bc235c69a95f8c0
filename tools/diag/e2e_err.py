import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from test_gpu_e2e import _build, _oracle_logits
from oracle import frontend as ofe, tcs as otcs
from oracle.primitives import bf16_round
from thunder_speech_amd import tensors as T
for rb in (1, 3):
    module, arch, sd, dsd = _build(rb)
    rng = np.random.Generator(np.random.PCG64(3))
    wav = torch.from_numpy((0.1 * rng.standard_normal((2, 40000))).astype(np.float32)); wav[1, 25000:] = 0
    lengths = torch.tensor([40000.0, 25000.0])
    # layer-by-layer comparison
    feats, fl = module.audio_transform(wav.cuda(), lengths.cuda())
    of, ofl = ofe.filterbank_features(wav, lengths)
    print(rb, "features max err", float((feats.float().cpu() - of).abs().max()), "vs bf16(of)", float((feats.float().cpu() - bf16_round(of)).abs().max()))
    x, l = feats, fl
    xe, le = bf16_round(of), ofl          # emulation fed by oracle features
    xh, lh = feats.float().cpu(), ofl     # emulation fed by HIP features
    x32, l32 = of, ofl
    for i, (blk, spec) in enumerate(zip(module.encoder, arch)):
        x, l = blk(x, l)
        xe, le = otcs.block_forward(spec, sd, f"{i}.", xe, le, emulate_bf16=True)
        xh, lh = otcs.block_forward(spec, sd, f"{i}.", xh, lh, emulate_bf16=True)
        x32, l32 = otcs.block_forward(spec, sd, f"{i}.", x32, l32)
        g = x.float().cpu()
        sc = float(x32.abs().max())
        print(f"  block {i:2d}: scale {sc:7.3f}  |hip-emu(hipfeat)| max {float((g-xh).abs().max()):.4f} rms {float((g-xh).pow(2).mean().sqrt()):.5f}   |hip-emu| max {float((g-xe).abs().max()):.4f}   |hip-fp32| max {float((g-x32).abs().max()):.4f} rms {float((g-x32).pow(2).mean().sqrt()):.5f}   |emu-fp32| rms {float((xe-x32).pow(2).mean().sqrt()):.5f}")
