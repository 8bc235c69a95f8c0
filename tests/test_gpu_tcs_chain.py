"""GPU parity of the chained split kernel (ts_tcs_chain_fwd): all repeats of a block in ONE persistent launch, tiles of repeat r + 1
released by per-(layer, clip, time tile) counters.  The chain must be BIT-IDENTICAL to the same repeats launched one by one
(ts_tcs_subblock_fwd: identical arithmetic, only the hand-over differs) and match the CPU oracle's block
(reference quartznet/blocks.py:317-338, citrinet/blocks.py:175-197)."""
import pytest
import torch

from oracle import tcs as otcs
from oracle.primitives import bf16_round

pytestmark = pytest.mark.gpu


def _block(spec, seed, citrinet=False):
    from thunder_speech_amd.citrinet.blocks import CitrinetBlock
    from thunder_speech_amd.quartznet.blocks import QuartznetBlock
    sd = {k[2:]: v for k, v in otcs.synth_encoder_state([spec], seed=seed, calibrate=True).items()}
    cls = CitrinetBlock if citrinet else QuartznetBlock
    blk = cls(spec.in_ch, spec.out_ch, repeat=spec.repeat, kernel_size=(spec.kernel,), stride=(spec.stride,), dilation=(spec.dilation,),
              residual=spec.residual, separable=True)
    blk.load_state_dict(sd, strict=True)
    return blk.cuda().eval(), sd


def _run(blk, x, lengths, chain, internal=True):
    from thunder_speech_amd import plan, tensors as TS
    plan.CHAIN = "force" if chain else False
    try:
        xi = TS.pack(x, lengths, slot=("chain-test", id(blk)))
        y, out_len, _ = blk._run_fused(xi, lengths, internal=internal)
        torch.cuda.synchronize()
        return y.clone(), out_len
    finally:
        plan.CHAIN = True


def _launch_counter():
    """Counts C-ABI launches by name, so that a test can tell that the chain entry point really ran."""
    from thunder_speech_amd import _lib
    L = _lib.lib()
    seen = {"chain": 0, "single": 0}
    orig_chain, orig_single = L.ts_tcs_chain_fwd, L.ts_tcs_subblock_fwd

    class Counting:
        def __init__(self, fn, key):
            self.fn, self.key = fn, key

        def __call__(self, *a):
            st = self.fn(*a)
            if st == 0:
                seen[self.key] += 1
            return st
    return L, seen, orig_chain, orig_single, Counting


@pytest.mark.parametrize("cin,cout,k,repeat,b,t,lens", [
    (256, 256, 33, 5, 3, 751, [751, 600, 13]),           # 192-frame x 256-channel tiles
    (256, 512, 51, 5, 4, 300, [300, 211, 300, 1]),       # first repeat 256 -> 512: per-layer stage counts differ
    (512, 512, 63, 5, 5, 403, [403, 402, 97, 96, 95]),   # lengths around the 96-frame tile edges
    (512, 512, 75, 3, 2, 200, [200, 150]),
    (128, 64, 11, 4, 2, 140, [140, 77]),
])
def test_chain_is_bit_identical_to_single_launches_and_matches_oracle(cin, cout, k, repeat, b, t, lens):
    from thunder_speech_amd import plan
    spec = otcs.BlockSpec(cin, cout, repeat=repeat, kernel=k, stride=1, dilation=1, residual=True, separable=True)
    blk, sd = _block(spec, seed=k)
    g = torch.Generator().manual_seed(k)
    x = bf16_round(torch.randn(b, cin, t, generator=g)).cuda()
    lengths = torch.tensor(lens).cuda()
    L, seen, oc, os_, Counting = _launch_counter()
    L.ts_tcs_chain_fwd, L.ts_tcs_subblock_fwd = Counting(oc, "chain"), Counting(os_, "single")
    try:
        one, _ = _run(blk, x, lengths, chain=False)
        assert seen == {"chain": 0, "single": repeat}
        got, out_len = _run(blk, x, lengths, chain=True)
        assert seen == {"chain": 1, "single": repeat}, seen          # the whole internal block was ONE launch
    finally:
        L.ts_tcs_chain_fwd, L.ts_tcs_subblock_fwd = oc, os_
    layers = blk._cache.get(blk._params(), blk._compile)
    assert plan.chain_status(layers) == 0
    assert torch.equal(got.view(torch.int16), one.view(torch.int16))
    want, want_len = otcs.block_forward(spec, sd, "", x.float().cpu(), lengths.cpu(), emulate_bf16=True)
    assert torch.equal(out_len.cpu(), want_len)
    gotc = got.float().cpu()
    scale = max(1.0, float(want.abs().max()))
    for i, n in enumerate(lens):
        assert float((gotc[i, :, :n] - want[i, :, :n]).abs().max()) <= 0.02 * scale
        assert float(gotc[i, :, n:].abs().max()) == 0.0 if n < t else True


def test_caller_visible_block_chains_all_but_the_last_repeat():
    """The last repeat of a caller-visible block keeps the reference's values beyond the length (quirk A2) and therefore runs on its
    own; the four repeats in front of it are one chain launch."""
    spec = otcs.BlockSpec(512, 512, repeat=5, kernel=63, stride=1, dilation=1, residual=True, separable=True)
    blk, sd = _block(spec, seed=3)
    x = bf16_round(torch.randn(2, 512, 300, generator=torch.Generator().manual_seed(3))).cuda()
    lengths = torch.tensor([300, 123]).cuda()
    L, seen, oc, os_, Counting = _launch_counter()
    L.ts_tcs_chain_fwd, L.ts_tcs_subblock_fwd = Counting(oc, "chain"), Counting(os_, "single")
    try:
        got, _ = _run(blk, x, lengths, chain=True, internal=False)
        assert seen == {"chain": 1, "single": 1}, seen
        one, _ = _run(blk, x, lengths, chain=False, internal=False)
    finally:
        L.ts_tcs_chain_fwd, L.ts_tcs_subblock_fwd = oc, os_
    assert torch.equal(got.view(torch.int16), one.view(torch.int16))
    want, _ = otcs.block_forward(spec, sd, "", x.float().cpu(), lengths.cpu(), emulate_bf16=True)
    assert float((got.float().cpu() - want).abs().max()) <= 0.02 * max(1.0, float(want.abs().max()))


def test_citrinet_block_chain_two_output_channel_splits():
    """1024 output channels = two 512-channel tiles per (clip, time tile): a tile of the next repeat waits for BOTH."""
    spec = otcs.BlockSpec(1024, 1024, repeat=5, kernel=11, stride=1, dilation=1, residual=True, separable=True, family="citrinet")
    blk, sd = _block(spec, seed=11, citrinet=True)
    x = bf16_round(torch.randn(3, 1024, 260, generator=torch.Generator().manual_seed(11))).cuda()
    lengths = torch.tensor([260, 200, 31]).cuda()
    got, out_len = _run(blk, x, lengths, chain=True)
    one, _ = _run(blk, x, lengths, chain=False)
    assert torch.equal(got.view(torch.int16), one.view(torch.int16))
    want, want_len = otcs.block_forward(spec, sd, "", x.float().cpu(), lengths.cpu(), emulate_bf16=True)
    assert torch.equal(out_len.cpu(), want_len)
    gotc = got.float().cpu()
    scale = max(1.0, float(want.abs().max()))
    for i, n in enumerate(lengths.tolist()):
        assert float((gotc[i, :, :n] - want[i, :, :n]).abs().max()) <= 0.02 * scale


@pytest.mark.parametrize("c,k,b", [(512, 63, 64), (256, 33, 64), (512, 75, 40)])
def test_full_size_chain_repeated_under_load_stays_bit_identical(c, k, b):
    """C2 geometry (64 x 751 frames: two 96-frame tiles per compute unit and layer; 40 clips: an uneven tile count), the hand-over
    exercised 20 times back to back with a competing memory stream on a second HIP stream (uneven load is where a missing release /
    acquire shows, MI355X_MICROARCH.md 'Test every hand-off under UNEVEN load').  Every repetition must reproduce the
    one-launch-per-repeat result bit for bit, and no wait may have timed out."""
    from thunder_speech_amd import plan
    spec = otcs.BlockSpec(c, c, repeat=5, kernel=k, stride=1, dilation=1, residual=True, separable=True)
    blk, _ = _block(spec, seed=k + b)
    g = torch.Generator().manual_seed(b)
    x = bf16_round(torch.randn(b, c, 751, generator=g)).cuda()
    lens = [751 - 17 * (i % 23) for i in range(b)]
    lengths = torch.tensor(lens).cuda()
    one, _ = _run(blk, x, lengths, chain=False)
    noise = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    for rep in range(20):
        if rep % 2:
            with torch.cuda.stream(side):
                for _ in range(3):
                    noise.mul_(1.0001)
        got, _ = _run(blk, x, lengths, chain=True)
        assert torch.equal(got.view(torch.int16), one.view(torch.int16)), f"repetition {rep}"
    torch.cuda.synchronize()
    layers = blk._cache.get(blk._params(), blk._compile)
    assert plan.chain_status(layers) == 0
    assert torch.isfinite(one.float()).all()
