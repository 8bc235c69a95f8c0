"""Round-5 GPU tests, all through the C ABI: the squeeze-excite bottleneck kernels (forward and autograd backward) against float64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ptr(t):
    return t.data_ptr()


@pytest.mark.parametrize("b,c,r", [(32, 1024, 128), (3, 24, 3), (5, 200, 25), (1, 64, 8), (7, 1536, 192)])
def test_squeeze_excite_bottleneck_forward_and_backward_match_float64_autograd(b, c, r):
    """ts_train_se_gate_fwd / _bwd (citrinet/blocks.py:72-83: Linear -> ReLU -> Linear -> sigmoid, both bias-free) vs torch float64 autograd;
    ts_se_gate_fwd's inference path runs the same forward kernel."""
    from thunder_speech_amd import _lib
    L = _lib.lib()
    gen = torch.Generator().manual_seed(b * 1000 + c)
    mean = torch.randn(b, c, generator=gen)
    w1 = torch.randn(r, c, generator=gen) / c ** 0.5
    w2 = torch.randn(c, r, generator=gen) / r ** 0.5
    dg = torch.randn(b, c, generator=gen)
    m64, a64, b64 = mean.double().requires_grad_(True), w1.double().requires_grad_(True), w2.double().requires_grad_(True)
    h64 = torch.relu(m64 @ a64.t())
    g64 = torch.sigmoid(h64 @ b64.t())
    (g64 * dg.double()).sum().backward()

    d = lambda t: t.to(DEV).contiguous()
    mean_d, w1_d, w2_d, dg_d = d(mean), d(w1), d(w2), d(dg)
    f32 = dict(dtype=torch.float32, device=DEV)
    h, g = torch.full((b, r), float("nan"), **f32), torch.full((b, c), float("nan"), **f32)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.ts_train_se_gate_fwd(_ptr(mean_d), _ptr(w1_d), _ptr(w2_d), _ptr(h), _ptr(g), b, c, r, st), "fwd")
    np.testing.assert_allclose(h.cpu().numpy(), h64.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(g.cpu().numpy(), g64.detach().numpy(), atol=2e-5)
    g2 = torch.full((b, c), float("nan"), **f32)                  # hid == NULL (the inference call)
    _lib.check(L.ts_train_se_gate_fwd(_ptr(mean_d), _ptr(w1_d), _ptr(w2_d), None, _ptr(g2), b, c, r, st), "fwd")
    assert torch.equal(g, g2)

    nan = lambda *s: torch.full(s, float("nan"), **f32)
    dz, dh, dmean, dw1, dw2 = nan(b, c), nan(b, r), nan(b, c), nan(r, c), nan(c, r)
    _lib.check(L.ts_train_se_gate_bwd(_ptr(dg_d), _ptr(g), _ptr(h), _ptr(mean_d), _ptr(w1_d), _ptr(w2_d), _ptr(dz), _ptr(dh), _ptr(dmean),
                                      _ptr(dw1), _ptr(dw2), b, c, r, st), "bwd")
    for got, want in ((dmean, m64.grad), (dw1, a64.grad), (dw2, b64.grad)):
        s = max(float(want.abs().max()), 1e-6)
        assert float((got.cpu().double() - want).abs().max()) <= 2e-5 * s + 1e-7


def test_squeeze_excite_train_node_uses_no_aten_matmul():
    """SqueezeExciteTrain (train_ops.py) end to end vs float64 autograd, with torch's matmul poisoned: the bottleneck runs on the library's kernels."""
    from thunder_speech_amd import train_ops as T
    gen = torch.Generator().manual_seed(3)
    b, c, r, t = 4, 64, 8, 53
    x = torch.randn(b, c, t, generator=gen)
    w1, w2 = torch.randn(r, c, generator=gen) / 8, torch.randn(c, r, generator=gen) / 3
    cot = torch.randn(b, c, t, generator=gen)
    x64, a64, b64 = x.double().requires_grad_(True), w1.double().requires_grad_(True), w2.double().requires_grad_(True)
    gate = torch.sigmoid(torch.relu(x64.mean(-1) @ a64.t()) @ b64.t())
    ((x64 * gate[:, :, None]) * cot.double()).sum().backward()

    xd = x.to(DEV).requires_grad_(True)
    w1d, w2d = w1.to(DEV).requires_grad_(True), w2.to(DEV).requires_grad_(True)
    real = torch.Tensor.__matmul__
    try:
        torch.Tensor.__matmul__ = lambda *a: (_ for _ in ()).throw(AssertionError("ATen matmul on the squeeze-excite path"))
        y = T.SqueezeExciteTrain.apply(xd, w1d, w2d)
        (y.float() * cot.to(DEV)).sum().backward()
    finally:
        torch.Tensor.__matmul__ = real
    for got, want in ((xd.grad, x64.grad), (w1d.grad, a64.grad), (w2d.grad, b64.grad)):
        s = max(float(want.abs().max()), 1e-6)
        assert float((got.cpu().double() - want).abs().max()) <= 1e-4 * s
