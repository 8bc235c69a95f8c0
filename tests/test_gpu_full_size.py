"""Full-size oracle comparisons for the BASELINE.json configurations whose pytest cases are small: the tile schedules
of 32 x 20 s Citrinet-1024 (C3), 32 x 10 s QuartzNet15x5 training (C4, one rank's share of global 256) and 16 x 20 s wav2vec2-large (C5) run at the
sizes bench.py measures, and clip 0 (C4: the loss and the gradients nearest to it) is compared with the CPU oracle.  Marked `slow` (a minute
or two of oracle time on the box's host cores each); they still run under plain `-m gpu`."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import frontend as ofe
from oracle import tcs as otcs

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _threads():
    torch.set_num_threads(min(16, os.cpu_count() or 1))


def test_c3_citrinet1024_32x20s_clip0_matches_oracle():
    """BASELINE.json configs[2] at full size: 32 x 20 s through the 23-block Citrinet-1024 stack (1024-channel tiles in two output-channel
    splits, squeeze-excite over 2001 / 1001 / 501 / 251 frames, strided residuals), logits of clip 0 on all 251 frames vs the fp32 oracle.
    Stated tolerance of the bf16 path: 2 % of the logit scale (measured 0.4 %), rms 0.5 %."""
    from thunder_speech_amd.citrinet.compatibility import CITRINET_1024_KERNELS, CITRINET_1024_STRIDES, build_synthetic_citrinet
    from thunder_speech_amd.utils import variance_preserving_init_
    torch.manual_seed(0)
    module = build_synthetic_citrinet()
    variance_preserving_init_(module.encoder, module.decoder, seed=0)
    module = module.cuda().eval()
    wav = 0.1 * torch.randn(32, 16000 * 20, generator=torch.Generator().manual_seed(1234))
    lengths = torch.full((32,), 16000 * 20, dtype=torch.int32)
    with torch.no_grad():
        logits, out_len = module(wav.cuda(), lengths.cuda())
        torch.cuda.synchronize()
    assert torch.isfinite(logits).all() and int(out_len[0]) == 251
    _threads()
    arch = otcs.citrinet_arch([1024] * 21, CITRINET_1024_KERNELS, CITRINET_1024_STRIDES, feat_in=80)
    sd = {k: v.detach().cpu() for k, v in module.encoder.state_dict().items()}
    dsd = {k: v.detach().cpu() for k, v in module.decoder.state_dict().items()}
    with torch.no_grad():
        feats, fl = ofe.filterbank_features(wav[:1], lengths[:1], ofe.FrontendConfig(n_window_size=400, nfilt=80))
        enc, _ = otcs.encoder_forward(arch, sd, feats, fl)
        ref = otcs.conv1d_decoder_forward(dsd, enc)
    got = logits[:1].float().cpu()
    scale = float(ref.abs().max())
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) <= 0.02 * scale and float((got - ref).pow(2).mean().sqrt()) <= 0.005 * scale
    # a clip's logits do not depend on its batch mates: the last clip alone must reproduce its row of the batch bit for bit
    with torch.no_grad():
        alone, _ = module(wav[31:].cuda(), lengths[31:].cuda())
    assert torch.equal(alone[0], logits[31])


def test_c5_wav2vec2_large_16x20s_clip0_matches_oracle():
    """BASELINE.json configs[4] at full size: 16 x 20 s (999 frames per clip: 8 query tiles of the fused attention per head, 16 clips in grid.y
    of every GEMM), last_hidden_state of clip 0 vs the fp32 oracle.  Outputs are LayerNorm-ed (unit scale): max 0.1, rms 0.015 (measured 0.040 /
    0.0086 after 24 layers of bf16 operands)."""
    sys.path.insert(0, ROOT)
    from tools.bench_c5 import config, random_state
    from oracle import w2v as ow
    from thunder_speech_amd.huggingface.encoder import Wav2Vec2Plan
    from thunder_speech_amd.huggingface.transform import Wav2Vec2Preprocess
    cfg = config(False, 24)
    sd = random_state(cfg)
    plan = Wav2Vec2Plan(cfg, sd, "cuda", precision="bf16")
    x = 0.1 * torch.randn(16, 16000 * 20, generator=torch.Generator().manual_seed(1234))
    lengths = torch.full((16,), 16000 * 20, dtype=torch.int32)
    with torch.no_grad():
        xn, _ = Wav2Vec2Preprocess()(x.cuda(), lengths.cuda())
        out = plan.forward(xn, None)
        torch.cuda.synchronize()
    assert out.shape == (16, 999, 1024) and torch.isfinite(out).all()
    _threads()
    ocfg = ow.W2VConfig(conv_dim=cfg.conv_dim, conv_kernel=cfg.conv_kernel, conv_stride=cfg.conv_stride, hidden_size=cfg.hidden_size,
                        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                        intermediate_size=cfg.intermediate_size, num_conv_pos_embeddings=cfg.num_conv_pos_embeddings,
                        num_conv_pos_embedding_groups=cfg.num_conv_pos_embedding_groups)
    x0 = x[:1]
    xr = (x0 - x0.mean(dim=1, keepdim=True)) / torch.sqrt(x0.var(dim=1, keepdim=True) + 1e-7)
    with torch.no_grad():
        ref, _ = ow.forward(ocfg, sd, xr)
    err = (out[:1].float().cpu() - ref).abs()
    assert float(err.max()) <= 0.1 and float(err.pow(2).mean().sqrt()) <= 0.015, (float(err.max()), float(err.pow(2).mean().sqrt()))


# bf16 rows: measured 6.4e-4 / 1.7e-2 / 6.7e-2 on this case (round 5); the tolerances are ~3x that
BF16_LOSS_TOL, BF16_DEC_TOL, BF16_BLK_TOL = 3e-3, 5e-2, 2e-1
# tolerances: (loss, relative), (decoder gradients, relative L2), (last block's gradients, relative L2)
C4_TOL = {"fp32": (1e-4, 2e-3, 1e-2), "bf16": (BF16_LOSS_TOL, BF16_DEC_TOL, BF16_BLK_TOL)}


@pytest.mark.parametrize("act", ["fp32", "bf16"])
def test_c4_quartznet15x5_local32x10s_training_step_matches_oracle_autograd(act):
    """BASELINE.json configs[3], one rank's share at 8 GPUs: local batch 32 x 10 s, QuartzNet15x5 in train mode (batch-statistics BatchNorm over
    32 x 501 frames, quirk A4).  "fp32": f32 activations (the reference's arithmetic); "bf16": the MEASURED mode of bench.py's c4_phase2 / c4_ddp
    (bf16 activation rows, f32 master weights, gradients and statistics -- the reference under Lightning's bf16-mixed precision).  The CTC loss of one
    training_step and the gradients nearest to it -- the decoder's and the last block's -- vs torch autograd through the fp32 oracle fed the HIP
    front end's features (the front end has its own parity tests).  f32: loss 1e-4 relative; decoder gradients 2e-3, last block 1e-2 in the
    relative L2 norm (the tolerances of the small case, tests/test_gpu_configs.py).  bf16: C4_TOL, about 3x what 17 blocks of bf16 storage were
    measured to cost on this case.  Every one of the 356 encoder gradients must exist and be finite."""
    from thunder_speech_amd import train_ops
    from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
    train_ops.set_activation_dtype(act)
    try:
        _c4_case(act, build_synthetic_quartznet)
    finally:
        train_ops.set_activation_dtype("fp32")


def _c4_case(act, build_synthetic_quartznet):
    tol_loss, tol_dec, tol_blk = C4_TOL[act]
    arch = otcs.quartznet_arch(repeat_blocks=3)
    sd = otcs.synth_encoder_state(arch, seed=0, calibrate=True, main_gamma=0.3)
    dsd = otcs.synth_decoder_state(1024, 29, seed=1)
    m = build_synthetic_quartznet(repeat_blocks=3, encoder_state=sd, decoder_state=dsd).cuda().train()
    m.audio_transform[0].layer[0].dither = 0.0
    g = torch.Generator().manual_seed(1234)
    wav = 0.1 * torch.randn(32, 16000 * 10, generator=g)
    lens = [160000 - 4000 * (i % 9) for i in range(32)]
    for i, n in enumerate(lens):
        wav[i, n:] = 0
    lengths = torch.tensor([float(n) for n in lens])
    texts = ["".join(chr(97 + int(c)) for c in torch.randint(0, 26, (int(n),), generator=g)) for n in torch.randint(60, 140, (32,), generator=g)]
    loss = m.training_step((wav.cuda(), lengths.cuda(), texts), 0)
    loss.backward()
    with torch.no_grad():
        feats, fl = m.audio_transform(wav.cuda(), lengths.cuda())
    torch.cuda.synchronize()
    _threads()
    # the oracle needs gradients only for the parameters compared: the decoder and block 17 (everything upstream is a constant input to them)
    x, xl = feats.float().cpu(), fl.cpu()
    with torch.no_grad():
        for i, spec in enumerate(arch[:-1]):
            x, xl = otcs.block_forward(spec, sd, f"{i}.", x, xl, training=True)
    keys = [k for k in sd if k.startswith("17.") and sd[k].is_floating_point() and "running" not in k]
    sd_ref = {k: (v.clone().requires_grad_(True) if k in keys else v.clone()) for k, v in sd.items()}
    dref = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    x, xl = otcs.block_forward(arch[-1], sd_ref, "17.", x, xl, training=True)
    logits = otcs.conv1d_decoder_forward(dref, x)
    y, yl = m.text_transform.encode(texts)
    ref = torch.nn.functional.ctc_loss(logits.permute(2, 0, 1).log_softmax(2), y, xl.long(), yl, blank=m.text_transform.vocab.blank_idx,
                                       reduction="mean", zero_infinity=True)
    ref.backward()
    params = dict(m.encoder.named_parameters())
    e_loss = abs(float(loss) - float(ref)) / max(1.0, abs(float(ref)))
    e_dec = max(float((p.grad.cpu() - dref[k].grad).norm()) / float(dref[k].grad.norm()) for k, p in m.decoder.named_parameters())
    e_blk = max(float((params[k].grad.cpu() - sd_ref[k].grad).norm()) / float(sd_ref[k].grad.norm()) for k in keys)
    print(f"C4 local 32 x 10 s, {act} activations: loss error {e_loss:.2e} (relative), decoder gradients {e_dec:.2e}, block 17 gradients {e_blk:.2e} (relative L2)")
    assert e_loss <= tol_loss, (float(loss), float(ref))
    assert len(params) == 356 and all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in params.values())
    assert e_dec <= tol_dec and e_blk <= tol_blk, (e_dec, e_blk)


def test_c2_quartznet15x5_64x15s_first_clips_match_oracle():
    """BASELINE.json configs[1] at full size, the headline's own model (bench.build_model: variance-preserving random weights) and batch
    (64 x 15 s): logits of the first four clips on all 751 frames vs the fp32 oracle -- max error 2 % of the logit scale, rms 0.5 % (measured 1.0 /
    0.23 %) -- and the device's greedy path against it: the argmax agrees on EVERY frame the oracle decides by more than six sigma of the measured
    error, and after dropping the undecided frames the collapsed sequences (what predict() turns into text, text_processing/transform.py:
    107-110) are identical.  Strict all-frame identity is not assertable with random weights at this depth (tools/diag/margin_study.py: the
    frames of a clip are not linearly separable above the bf16 deviation); the margin-calibrated QuartzNet5x5 fixture asserts it for the
    shallower model (tests/test_gpu_configs.py)."""
    sys.path.insert(0, ROOT)
    import bench
    from oracle import decode as odec
    from thunder_speech_amd.module import greedy_decode
    module = bench.build_model(torch.device("cuda", 0))
    wav = 0.1 * torch.randn(64, 16000 * 15, generator=torch.Generator().manual_seed(1234))
    lengths = torch.full((64,), 16000 * 15, dtype=torch.int32)
    with torch.no_grad():
        logits, _ = module(wav.cuda(), lengths.cuda())
        ids, collapsed, counts = greedy_decode(logits)
        torch.cuda.synchronize()
    assert logits.shape == (64, 29, 751) and torch.isfinite(logits).all()
    _threads()
    n = 4
    arch = otcs.quartznet_arch(repeat_blocks=3)
    sd = {k: v.detach().cpu() for k, v in module.encoder.state_dict().items()}
    dsd = {k: v.detach().cpu() for k, v in module.decoder.state_dict().items()}
    with torch.no_grad():
        feats, fl = ofe.filterbank_features(wav[:n], lengths[:n])
        enc, _ = otcs.encoder_forward(arch, sd, feats, fl)
        ref = otcs.conv1d_decoder_forward(dsd, enc).numpy()
    got = logits[:n].float().cpu().numpy()
    scale = float(np.abs(ref).max())
    err = got - ref
    rms = float(np.sqrt(np.mean(err.astype(np.float64) ** 2)))
    assert float(np.abs(err).max()) <= 0.02 * scale and rms <= 0.005 * scale, (float(np.abs(err).max()) / scale, rms / scale)
    top2 = np.sort(ref, axis=1)[:, -2:, :]
    decided = (top2[:, 1] - top2[:, 0]) > 6 * rms
    a_got, a_ref = got.argmax(1), ref.argmax(1)
    assert decided.mean() > 0.9
    assert np.array_equal(ids[:n].cpu().numpy(), a_got)                       # device argmax == host argmax of the device logits
    assert np.array_equal(a_got[decided], a_ref[decided])
    for b in range(n):
        assert list(odec.collapse_repeats(a_got[b][decided[b]])) == list(odec.collapse_repeats(a_ref[b][decided[b]]))
        assert collapsed[b, : int(counts[b])].cpu().tolist() == list(odec.collapse_repeats(a_got[b]))   # device collapse == host collapse
