"""GPU parity of the whole path: waveform -> FilterbankFeatures -> QuartznetEncoder -> decoder -> greedy decode."""
import numpy as np
import pytest
import torch

from oracle import decode as odec
from oracle import frontend as ofe
from oracle import tcs as otcs
from oracle.primitives import bf16_round

pytestmark = pytest.mark.gpu
LABELS = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]


def _build(repeat_blocks, enc_seed=0, dec_seed=1):
    from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
    arch = otcs.quartznet_arch(repeat_blocks=repeat_blocks)
    sd = otcs.synth_encoder_state(arch, seed=enc_seed, calibrate=True)
    dsd = otcs.synth_decoder_state(1024, 29, seed=dec_seed, gain=4.0)
    module = build_synthetic_quartznet(repeat_blocks=repeat_blocks, encoder_state=sd, decoder_state=dsd).cuda().eval()
    return module, arch, sd, dsd


def _oracle_logits(arch, sd, dsd, wav, lengths, emulate):
    feats, fl = ofe.filterbank_features(wav, lengths)
    enc, el = otcs.encoder_forward(arch, sd, bf16_round(feats) if emulate else feats, fl, emulate_bf16=emulate)
    return otcs.conv1d_decoder_forward(dsd, enc, emulate_bf16=emulate), el


def _rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, dtype=np.float64)))))


def test_quartznet5x5_logits_match_reference_fixture(golden):
    """Fixture = logits of the REAL reference modules (fp32).  Stated tolerance of the bf16 path:
    5e-2 of the logit scale; and the HIP path must be no less accurate than an fp32-accumulating bf16 evaluation
    of the same network (oracle emulate_bf16)."""
    g = golden("qn5x5_e2e.npz")
    module, arch, sd, dsd = _build(1, int(g["enc_seed"]), int(g["dec_seed"]))
    rng = np.random.Generator(np.random.PCG64(int(g["wav_seed"])))
    wav = torch.from_numpy((0.1 * rng.standard_normal(tuple(g["wav_shape"]))).astype(np.float32))
    for b, z in enumerate(g["wav_zero_from"]):
        wav[b, int(z):] = 0
    lengths = torch.from_numpy(g["wav_lengths"])
    logits, out_len = module(wav.cuda(), lengths.cuda())
    torch.cuda.synchronize()
    assert np.array_equal(out_len.cpu().numpy(), g["out_lengths"])
    got = logits.float().cpu().numpy()
    ref = g["logits"]
    emu, _ = _oracle_logits(arch, sd, dsd, wav, lengths, True)
    scale = float(np.abs(ref).max())
    assert np.abs(got - ref).max() <= 0.05 * scale
    assert _rms(got - ref) <= 1.25 * _rms(emu.numpy() - ref) + 1e-3 * scale
    # greedy frames: every frame whose fp32 top-1/top-2 margin exceeds the tolerance must agree
    top2 = np.sort(ref, axis=1)[:, -2:, :]
    decided = (top2[:, 1] - top2[:, 0]) > 0.1 * scale
    assert decided.mean() > 0.5
    assert np.array_equal(got.argmax(1)[decided], ref.argmax(1)[decided])


@pytest.mark.parametrize("repeat_blocks", [1, 3])
def test_every_block_matches_oracle_teacher_forced(repeat_blocks):
    """Each QuartznetBlock fed the oracle's (bf16-rounded) activations must reproduce the oracle's bf16-ordered
    evaluation of that block to ~1 bf16 ulp -- the depth-independent statement of kernel parity (end-to-end
    differences in a random-weight 18-block stack are dominated by bf16 rounding noise)."""
    module, arch, sd, dsd = _build(repeat_blocks)
    g = torch.Generator().manual_seed(5)
    x = bf16_round(torch.randn(3, 64, 403, generator=g))
    lengths = torch.tensor([403, 300, 77])
    for i, (blk, spec) in enumerate(zip(module.encoder, arch)):
        want, want_len = otcs.block_forward(spec, sd, f"{i}.", x, lengths, emulate_bf16=True)
        got, got_len = blk(x.cuda(), lengths.cuda())
        assert torch.equal(got_len.cpu(), want_len)
        got = got.float().cpu()
        scale = max(1.0, float(want.abs().max()))
        err = (got - want).abs()
        assert float(err.max()) <= 0.016 * scale, f"block {i}: max err {float(err.max())} scale {scale}"
        assert _rms(err.numpy()) <= 1e-3 * scale, f"block {i}: rms err {_rms(err.numpy())} scale {scale}"
        x, lengths = want, want_len


def test_predict_strings_match_oracle():
    module, arch, sd, dsd = _build(1)
    rng = np.random.Generator(np.random.PCG64(77))
    wav = torch.from_numpy((0.1 * rng.standard_normal((3, 32000))).astype(np.float32))
    strings = module.predict(wav.cuda())
    lengths = torch.full((3,), 32000)
    ref, _ = _oracle_logits(arch, sd, dsd, wav, lengths, False)
    logits, _ = module(wav.cuda(), lengths.cuda())
    got = logits.float().cpu().numpy()
    vocab = odec.Vocab(list(LABELS))
    # identical to decoding our own logits on the host with the reference algorithm ...
    assert strings == odec.decode_prediction(odec.argmax_classes(got), vocab)
    # ... and to the fp32 oracle wherever its decision is not a near-tie (margin > 6 sigma of the bf16 noise)
    r = ref.numpy()
    noise = _rms(got - r)
    top2 = np.sort(r, axis=1)[:, -2:, :]
    decided = (top2[:, 1] - top2[:, 0]) > 6 * noise
    assert decided.mean() > 0.5
    assert np.array_equal(got.argmax(1)[decided], r.argmax(1)[decided])


def test_quartznet15x5_ragged_batch_matches_oracle():
    module, arch, sd, dsd = _build(3)
    rng = np.random.Generator(np.random.PCG64(3))
    wav = torch.from_numpy((0.1 * rng.standard_normal((2, 40000))).astype(np.float32))
    wav[1, 25000:] = 0
    lengths = torch.tensor([40000.0, 25000.0])          # float lengths as asr_collate emits (A5)
    logits, out_len = module(wav.cuda(), lengths.cuda())
    emu, el = _oracle_logits(arch, sd, dsd, wav, lengths, True)
    ref, _ = _oracle_logits(arch, sd, dsd, wav, lengths, False)
    assert torch.equal(out_len.cpu(), el) and out_len.dtype == el.dtype
    got = logits.float().cpu().numpy()
    # ALL frames are compared, incl. those beyond out_lengths (A2: predict() reads them).  The bf16 evaluation of
    # this random-weight 18-block stack deviates from fp32 by design (rounding noise amplified by depth); the HIP
    # path must not deviate more than the oracle's own bf16-ordered evaluation does.
    assert _rms(got - ref.numpy()) <= 1.25 * _rms(emu.numpy() - ref.numpy())
    assert np.isfinite(got).all()


def test_cpu_tensors_fail_loudly():
    module, *_ = _build(1)
    with pytest.raises(RuntimeError, match="no CPU"):
        module.cpu().predict(torch.zeros(1, 16000))


def test_hipgraph_replays_are_bit_identical_to_eager():
    """The bench replays the whole step from a hipGraph back to back: every replay must reproduce the eager logits
    bit for bit (no launch-order or cache-state dependence anywhere in the inference path)."""
    from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
    from thunder_speech_amd.utils import variance_preserving_init_
    module = build_synthetic_quartznet(repeat_blocks=3)
    variance_preserving_init_(module.encoder, module.decoder, seed=4)
    module = module.cuda().eval()
    g = torch.Generator().manual_seed(8)
    wav = (0.1 * torch.randn(8, 16000 * 6, generator=g)).cuda()
    lengths = torch.tensor([96000, 96000, 90000, 80000, 64000, 50000, 33000, 16000], dtype=torch.int32).cuda()
    with torch.no_grad():
        for _ in range(2):
            eager, eager_len = module(wav, lengths)
        eager = eager.clone()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                out, out_len = module(wav, lengths)
        for it in range(25):
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, eager), f"replay {it}: max diff {float((out - eager).abs().max())}"
    assert torch.equal(out_len, eager_len)


@pytest.mark.parametrize("cin,cout,k,repeat,t,lens,residual", [
    (512, 512, 63, 3, 751, [751, 700, 96, 1], True),          # the C2 body shape: 8 stages, rows requested two stages ahead
    (128, 640, 75, 4, 389, [389, 388, 97], False),            # 2 stages per tile, 640 output channels = two channel splits
    (64, 512, 33, 4, 192, [192, 191, 5], True),               # ONE 64-channel stage per tile: the single-set producer path
    (192, 256, 21, 2, 97, [97, 96], True),                    # odd stage count, a tile boundary one frame in
    (256, 1024, 83, 3, 193, [193, 100], True),                # longest taps the split kernel takes
    (512, 128, 5, 1, 1001, [1001, 640], True),                # shortest taps, single repeat with residual
])
def test_split_kernel_block_geometries_match_oracle(cin, cout, k, repeat, t, lens, residual):
    """Blocks whose sub-blocks all take the split kernel (c_in % 64 == 0, internal tail-zero tensors), over stage counts, channel splits, tap
    lengths and frame counts around the 96 / 192-frame tiles, against the oracle's bf16-ordered evaluation; a second call (warm arena
    buffers) must reproduce the first bit for bit.  (tools/diag/split_sweep.py draws such cases at random.)"""
    from thunder_speech_amd.quartznet.blocks import QuartznetBlock
    spec = otcs.BlockSpec(cin, cout, repeat=repeat, kernel=k, stride=1, dilation=1, residual=residual, separable=True)
    sd = {key[2:]: v for key, v in otcs.synth_encoder_state([spec], seed=k + t).items()}
    blk = QuartznetBlock(cin, cout, repeat=repeat, kernel_size=(k,), residual=residual, separable=True)
    blk.load_state_dict(sd, strict=True)
    blk = blk.cuda().eval()
    x = bf16_round(torch.randn(len(lens), cin, t, generator=torch.Generator().manual_seed(t)))
    lengths = torch.tensor(lens)
    want, want_len = otcs.block_forward(spec, sd, "", x, lengths, emulate_bf16=True)
    with torch.no_grad():
        got, got_len = blk(x.cuda(), lengths.cuda())
        again, _ = blk(x.cuda(), lengths.cuda())
    assert torch.equal(got_len.cpu(), want_len) and torch.equal(again, got)
    scale = max(1.0, float(want.abs().max()))
    assert float((got.float().cpu() - want).abs().max()) <= 0.016 * scale
