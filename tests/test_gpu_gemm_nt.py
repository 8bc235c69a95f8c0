"""GPU parity of the token-major bf16 GEMM with fused epilogue (csrc/gemm_nt.hip, ts_gemm_nt_bf16) -- what the wav2vec2 encoder's
linears and conv layers 1-6 run in bf16 mode -- against a float64 product of the same bf16 operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(x, w, bias, res, gelu, want32, want16):
    from thunder_speech_amd import _lib
    m, k = x.shape
    n = w.shape[0]
    y = torch.empty(m, n, dtype=torch.float32, device="cuda") if want32 else None
    y16 = torch.empty(m, n, dtype=torch.bfloat16, device="cuda") if want16 else None
    st = _lib.lib().ts_gemm_nt_bf16(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), bias.data_ptr() if bias is not None else None,
                                    res.data_ptr() if res is not None else None, res.stride(0) if res is not None else 0,
                                    y.data_ptr() if want32 else None, n, y16.data_ptr() if want16 else None, n, m, n, k, int(gelu),
                                    torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "ts_gemm_nt_bf16")
    return y, y16


@pytest.mark.parametrize("m,n,k,gelu,use_bias,use_res", [
    (256, 256, 32, False, False, False),          # one tile, one half-stage
    (300, 96, 64, False, True, True),             # ragged rows, partial column tile
    (999, 1024, 1024, True, True, False),         # q / k / v sized, GELU
    (1998, 1024, 4096, False, False, True),       # feed-forward output with the residual stream
    (513, 512, 1536, True, True, False),          # conv layer (k = 3 x 512)
    (257, 4096, 1024, True, True, False),         # many column tiles
    (4000, 32, 96, False, True, False),           # narrow output, three half-stages
])
def test_gemm_nt_matches_float64(m, n, k, gelu, use_bias, use_res):
    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) / k ** 0.5).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda", generator=g) if use_bias else None
    res = torch.randn(m, n, device="cuda", generator=g) if use_res else None
    ref = x.double() @ w.double().t()
    if bias is not None:
        ref = ref + bias.double()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    if res is not None:
        ref = ref + res.double()
    scale = max(1.0, float(ref.abs().max()))
    y, y16 = _run(x, w, bias, res, gelu, True, True)
    assert float((y.double() - ref).abs().max()) <= 2e-5 * scale * max(1.0, k / 1024)       # f32 accumulation of bf16 products
    assert float((y16.double() - ref).abs().max()) <= 2 ** -7 * scale                        # + one bf16 rounding
    y_only16 = _run(x, w, bias, res, gelu, False, True)[1]
    assert torch.equal(y_only16, y16)


def test_gemm_nt_overlapping_rows_and_in_place_residual():
    """The conv layers hand the GEMM rows that OVERLAP (row pitch stride * C < K = kernel * C); out_proj / output_dense accumulate
    into the residual stream in place (res == y)."""
    from thunder_speech_amd import _lib
    c, kern, stride, t_in = 64, 3, 2, 201
    t_out = (t_in - kern) // stride + 1
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(t_in, c, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(96, kern * c, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    y = torch.randn(t_out, 96, device="cuda", generator=g)
    ref = y.double() + torch.stack([x[stride * t: stride * t + kern].reshape(-1) for t in range(t_out)]).double() @ w.double().t()
    st = _lib.lib().ts_gemm_nt_bf16(x.data_ptr(), stride * c, w.data_ptr(), kern * c, None, y.data_ptr(), 96, y.data_ptr(), 96, None, 0, t_out,
                                    96, kern * c, 0, torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "ts_gemm_nt_bf16")
    assert float((y.double() - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


def test_gemm_nt_declines_shapes_it_does_not_take():
    from thunder_speech_amd import _lib
    x = torch.zeros(64, 40, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(48, 40, dtype=torch.bfloat16, device="cuda")
    y = torch.zeros(64, 48, device="cuda")
    st = _lib.lib().ts_gemm_nt_bf16(x.data_ptr(), 40, w.data_ptr(), 40, None, None, 0, y.data_ptr(), 48, None, 0, 64, 48, 40, 0,
                                    torch.cuda.current_stream().cuda_stream)
    assert st == _lib.TS_EUNSUPPORTED                                                       # n % 32, k % 32


@pytest.mark.parametrize("m,n,k,gelu,use_res", [(300, 96, 64, False, True), (999, 1024, 1024, True, False), (1998, 1024, 4096, False, True),
                                                 (513, 512, 1536, True, False), (257, 4096, 1024, True, False), (4000, 32, 96, False, False),
                                                 (700, 64, 32, False, False),
                                                 (9000, 2048, 256, True, True), (8200, 2080, 128, False, False), (15984, 3072, 1024, True, False)])
def test_gemm_nt_packed_weights_give_the_same_bits(m, n, k, gelu, use_res):
    """ts_gemm_nt_bf16_packed (B fragments from L2 straight into registers, ts_gemm_nt_pack_w) multiplies the same bf16 values in the
    same order as the LDS path: results are bit-identical for every epilogue form; the packed image itself is checked against the
    fragment layout it documents.  The last three shapes have more tiles than the chip has CUs (several rounds of workgroups,
    the blocked tile walk with short last bands), ragged in both directions."""
    from thunder_speech_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(7 * m + n + k)
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) / k ** 0.5).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda", generator=g)
    res = torch.randn(m, n, device="cuda", generator=g) if use_res else None
    L, stream = _lib.lib(), torch.cuda.current_stream().cuda_stream
    wf = torch.empty_like(w)
    _lib.check(L.ts_gemm_nt_pack_w(w.data_ptr(), k, n, k, wf.data_ptr(), stream), "ts_gemm_nt_pack_w")
    lane = torch.arange(64, device="cuda")
    want = w.view(n // 16, 16, k // 32, 4, 8)[:, lane & 15, :, lane >> 4, :].permute(1, 2, 0, 3)      # [n/16][k/32][lane][8]
    assert torch.equal(wf.view(n // 16, k // 32, 64, 8), want)
    for want32, want16 in ((True, True), (False, True), (True, False)):
        if not want32 and res is not None:
            continue
        y, y16 = _run(x, w, bias, res, gelu, want32, want16)
        yp = torch.empty_like(y) if want32 else None
        yp16 = torch.empty_like(y16) if want16 else None
        st = L.ts_gemm_nt_bf16_packed(x.data_ptr(), k, w.data_ptr(), k, wf.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else None,
                                      n if res is not None else 0, yp.data_ptr() if want32 else None, n, yp16.data_ptr() if want16 else None, n,
                                      m, n, k, int(gelu), stream)
        _lib.check(st, "ts_gemm_nt_bf16_packed")
        assert (not want32 or torch.equal(y, yp)) and (not want16 or torch.equal(y16, yp16))
