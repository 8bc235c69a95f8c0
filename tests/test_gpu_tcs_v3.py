"""GPU parity of the merged TCS kernel (csrc/tcs_v3.hip: depthwise as Toeplitz x time segments on v_mfma_f32_16x16x32_bf16) vs the CPU
oracle.  The kernel is opt-in (TS_TCS_V3=1: measured slower than the split kernel), so the tests switch it on."""
import os

import pytest
import torch

from oracle.primitives import bf16_round
from tests.test_gpu_tcs import _run_case_tail_zero

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _merged_kernel():
    old = os.environ.get("TS_TCS_V3")
    os.environ["TS_TCS_V3"] = "1"
    yield
    if old is None:
        os.environ.pop("TS_TCS_V3", None)
    else:
        os.environ["TS_TCS_V3"] = old


@pytest.mark.parametrize("cin,cout,k,t,lens,res", [
    (256, 256, 33, 751, [751, 600, 13], True),           # 2 chunks, residual stages
    (256, 256, 39, 300, [300, 299], False),
    (256, 512, 51, 300, [300, 211], True),               # 3 chunks, two output-channel splits
    (512, 512, 75, 200, [200, 1], True),                 # widest window (o = -40)
    (512, 512, 63, 751, [751, 640], False),
    (64, 64, 5, 128, [128, 100], False),                 # 1 chunk
    (320, 384, 11, 251, [251, 97], False),               # 5 stages, 384 output channels: partial split
    (128, 1024, 17, 150, [150, 64], False),              # four output-channel splits
    (256, 256, 33, 1400, [1400, 1399, 700, 5] * 40, True),          # more tiles than CUs: the stage stream runs across tiles
    (512, 512, 51, 570, [570, 569, 300] * 14 + [33], True),         # ragged last XCD range of the tile order
    (320, 512, 25, 300, [300, 150, 7] * 43 + [299], True),          # 5 + 5 stages per tile
])
def test_merged_kernel_matches_oracle(cin, cout, k, t, lens, res):
    _run_case_tail_zero(cin, cout, k, 1, 1, t, lens, res)


@pytest.mark.parametrize("k", [63, 39, 17])
def test_merged_kernel_back_to_back_launches_use_their_own_taps(k):
    """The tap image of stage i + 2 is fetched by DMA into the LDS the running iteration has just read: with L2-hot taps (the same
    layer again and again) a DMA that overtook the reads would show as an O(1) error.  Identity pointwise, distinct taps."""
    from thunder_speech_amd import plan, tensors as TS
    c, t, b = (512, 751, 2) if k == 63 else (256, 500, 3)
    pad = k // 2
    g = torch.Generator().manual_seed(k)
    dw = bf16_round(torch.randn(c, 1, k, generator=g) * 0.2)
    bn = [torch.ones(c), torch.zeros(c), torch.zeros(c), torch.ones(c) - 1e-3]
    layer = plan.make_tcs_layer("cuda", dw_w=dw, pw_w=torch.eye(c).reshape(c, c, 1), bn=bn, kernel=k, stride=1, dilation=1,
                                padding=pad, relu=False)
    x = bf16_round(torch.randn(b, c, t, generator=g))
    ref = torch.nn.functional.conv1d(x.double(), dw.double(), padding=pad, groups=c).float()
    li = torch.full((b,), t, dtype=torch.int32, device="cuda")
    xb = TS.backing(TS.pack(x.cuda(), li, slot=("v3b2b", k)))
    out = TS.arena(("v3b2bo", k), b, c, t, "cuda")
    scale = float(ref.abs().max())
    for it in range(40):
        y, _ = layer.run(xb, t, li, out=out, in_tail_zero=True, zero_tail=True)
        err = float((y[:, :, :t].float().cpu() - ref).abs().max())
        assert err <= 0.012 * scale, f"launch {it}: max err {err} (scale {scale})"
