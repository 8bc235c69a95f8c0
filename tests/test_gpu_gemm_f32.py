"""GPU parity of the f32 matrix-core GEMM (csrc/gemm_f32.hip, ts_gemm_f32) against a float64 product: every operand layout (N / T forms),
odd sizes, the outer contraction loop (the weight gradient's sum over clips), batches, beta, bias, bf16 operands and results."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(m, n, k, a_kc, b_kc, nkb=1, batch=1, beta=False, bias=False, in_bf16=False, out_bf16=False, seed=0):
    from thunder_speech_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(seed)
    dt = torch.bfloat16 if in_bf16 else torch.float32
    r4 = lambda x: (x + 3) // 4 * 4
    # A(z, j; m, k): stored [z][j][m][k_pad] (k contiguous) or [z][j][k][m_pad]
    a = torch.randn(batch, nkb, m, r4(k) + 4, generator=g) if a_kc else torch.randn(batch, nkb, k, r4(m) + 8, generator=g)
    b = torch.randn(batch, nkb, n, r4(k) + 4, generator=g) if b_kc else torch.randn(batch, nkb, k, r4(n) + 4, generator=g)
    a, b = a.to(dt).cuda(), b.to(dt).cuda()
    am = a[..., :k] if a_kc else a[..., :m].transpose(-1, -2)          # [z][j][m][k]
    bm = b[..., :k].transpose(-1, -2) if b_kc else b[..., :n]          # [z][j][k][n]
    want = torch.einsum("zjmk,zjkn->zmn", am.double(), bm.double())
    ldc = n + 5
    c0 = torch.randn(batch, m, ldc, generator=g).to(torch.bfloat16 if out_bf16 else torch.float32).cuda()
    c = c0.clone()
    bv = torch.randn(n, generator=g).cuda() if bias else None
    if beta:
        want = want + c0[..., :n].double()
    if bias:
        want = want + bv.double()
    st = L.ts_gemm_f32(a.data_ptr(), a.stride(2) if a_kc else 1, 1 if a_kc else a.stride(2), a.stride(0), a.stride(1),
                       b.data_ptr(), 1 if b_kc else b.stride(2), b.stride(2) if b_kc else 1, b.stride(0), b.stride(1),
                       c.data_ptr(), ldc, c.stride(0), bv.data_ptr() if bias else None, m, n, k, nkb, batch, int(in_bf16), int(out_bf16), int(beta),
                       torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "ts_gemm_f32")
    torch.cuda.synchronize()
    got = c[..., :n].double()
    scale = float(want.abs().max())
    tol = (8e-3 if out_bf16 else 2e-6 * max(1, (k * nkb) ** 0.5)) * scale + 1e-6
    assert float((got - want).abs().max()) <= tol, (float((got - want).abs().max()), tol)
    assert torch.equal(c[..., n:], c0[..., n:])                                   # nothing written beyond N


@pytest.mark.parametrize("a_kc", [True, False])
@pytest.mark.parametrize("b_kc", [True, False])
@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (300, 200, 70), (33, 17, 5), (512, 29, 1024), (1, 640, 129)])
def test_every_operand_layout_matches_float64(m, n, k, a_kc, b_kc):
    _run(m, n, k, a_kc, b_kc)


@pytest.mark.parametrize("kw", [dict(nkb=5), dict(batch=3), dict(beta=True), dict(bias=True), dict(in_bf16=True), dict(in_bf16=True, out_bf16=True),
                                dict(nkb=3, batch=2, beta=True, bias=True), dict(in_bf16=True, nkb=4, beta=True)])
def test_contraction_loop_batches_and_epilogue(kw):
    _run(200, 136, 52, True, False, **kw)
    _run(96, 260, 40, False, True, **kw)


def test_overlapping_rows_like_the_wav2vec2_conv_layers():
    """A(m, k) = x[m * lda + k] with K > lda: the im2col matrix of a strided conv over a time-major input IS the input (w2v_enc.hip)."""
    from thunder_speech_amd import _lib
    c_in, kernel, stride, t_in, c_out = 64, 3, 2, 201, 96
    t_out = (t_in - kernel) // stride + 1
    x = torch.randn(t_in, c_in, generator=torch.Generator().manual_seed(1)).cuda()
    w = torch.randn(c_out, kernel * c_in, generator=torch.Generator().manual_seed(2)).cuda()
    y = torch.empty(t_out, c_out, device="cuda")
    st = _lib.lib().ts_gemm_f32(x.data_ptr(), stride * c_in, 1, 0, 0, w.data_ptr(), 1, kernel * c_in, 0, 0, y.data_ptr(), c_out, 0, None, t_out, c_out,
                                kernel * c_in, 1, 1, 0, 0, 0, torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "ts_gemm_f32")
    cols = torch.stack([x[stride * t: stride * t + kernel].reshape(-1) for t in range(t_out)])
    want = cols.double() @ w.double().t()
    assert float((y.double() - want).abs().max()) <= 1e-4 * float(want.abs().max())


def test_unaligned_pitches_take_the_element_path():
    """Row pitches that are not multiples of 4 (attention scores [t][t] with t = 299): element-by-element loads, same result."""
    from thunder_speech_amd import _lib
    t, hd = 299, 48
    g = torch.Generator().manual_seed(3)
    p = torch.randn(t, t, generator=g).cuda()
    v = torch.randn(t, 3 * hd + 2, generator=g).cuda()
    out = torch.empty(t, hd + 1, device="cuda")
    st = _lib.lib().ts_gemm_f32(p.data_ptr(), t, 1, 0, 0, v.data_ptr(), v.stride(0), 1, 0, 0, out.data_ptr(), out.stride(0), 0, None, t, hd, t, 1, 1, 0, 0, 0,
                                torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "ts_gemm_f32")
    want = p.double() @ v[:, :hd].double()
    assert float((out[:, :hd].double() - want).abs().max()) <= 1e-4 * float(want.abs().max())
