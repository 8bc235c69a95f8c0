// Fused time-channel-separable sub-block for gfx950 (MI355X), inference.
//
//   y[b, co, t] = act( sum_ci Wf[co, ci] * mask(dw[b, ci, t]) + bias[co] + sum_cr Wr[co, cr] * mask(xres[b, cr, t*rs]) )
//   dw[b, ci, t] = sum_u taps[ci, u] * mask(x)[b, ci, t*stride + u*dil - pad]
//
// Replaces the reference's per-sub-block ATen chain (quartznet/blocks.py:166-182 masked_fill + conv1d
// (groups=C) + masked_fill + conv1d(k=1), :222 batch_norm, :332-337 residual add + relu).
//
// Design (see DESIGN.md "TCS kernel"):
//  * layout NCT-p: bf16 [B][C][Tp], time contiguous.  One workgroup (4 waves) owns a tile of TT output
//    frames x CO_WG output channels of one clip and loops over the input channels in chunks of 64.
//  * depthwise FIR on the matrix cores: v_mfma_f32_4x4x4_16b_bf16 computes 16 independent 4x4x4
//    products per instruction -- one block per channel.  For channel c the A block is a 4x4 slice of the
//    Toeplitz matrix of its taps (rows = 4 consecutive output frames; pre-shifted per row on the host),
//    the B block holds 4 consecutive input samples for each of 4 time runs.  A lane keeps its sliding
//    input window in registers, so each input sample is read from LDS ~once per 3 k-steps.
//    4.1x the fp32-VALU FMA rate measured on MI355X (240 vs 58 TMAC/s), fp32 accumulate.
//  * the depthwise result goes to LDS as [ci][t] bf16 (XOR-swizzled 16-B chunks) and is consumed as the
//    A operand of v_mfma_f32_32x32x16_bf16 through ds_read_b64_tr_b16 (hardware transpose read);
//    the B operand (BN-folded pointwise weights) is pre-packed per lane on the host and streamed from
//    L2 with one coalesced 1-KiB load per fragment.
//  * residual 1x1 conv = a second pass over the block input with the depthwise stage replaced by a
//    masked copy, accumulating into the same registers; bias + ReLU + bf16 pack in the epilogue.
#include "ts_common.hpp"

namespace ts {

constexpr int KC = 64;         // input channels per chunk
constexpr int NKP = 3;         // depthwise k-steps (of 4 samples) per pass

struct TcsArgs {
  const unsigned short* x;     // [B][c_in][pitch_in]
  const unsigned short* xres;  // [B][c_res][pitch_res]
  void* y;                     // [B][c_out][pitch_out] bf16 or f32
  const int* len_in;
  const int* len_res;
  const unsigned short* taps;  // [c_in_pad][4][4*nk]
  const unsigned short* pw_w;  // fragments
  const unsigned short* res_w;
  const float* bias;
  int c_in, c_out, c_res;
  int pitch_in, pitch_out, pitch_res;
  int t_out;
  int kernel, stride, dilation, padding;
  int npass;                   // nk = 3 * npass
  int woff;                    // padL8 - padL4: element offset of the lane windows inside an xs row
  int padl8;                   // xs row starts at input frame t0*stride - padl8
  int xe;                      // staged elements per xs row (multiple of 8)
  int xpitch;                  // xs row pitch in elements (multiple of 4)
  int relu;
  int res_stride;
  int kt_main, kt_res;         // k-steps (16 channels) in the packed weights = c_pad64 / 16
};

// [ci][t] bf16 tile of the depthwise output / identity input, 16-byte chunks XOR-swizzled so that both
// the 8-byte row writes and the transposed reads spread over the banks.
template <int TT>
struct DwTile {
  static constexpr int ROWB = TT * 2;
  __device__ static __forceinline__ int sw(int c) {
    return TT == 128 ? (c & 3) * 5 : ((((c >> 1) & 1) << 2) | (c & 3));
  }
  __device__ static __forceinline__ int addr(int c, int t) {
    return c * ROWB + ((((t >> 3) ^ sw(c))) << 4) + ((t & 7) << 1);
  }
};

__device__ __forceinline__ int conv_len(int len, int k, int s, int p, int d) {
  const int num = len + 2 * p - d * (k - 1) - 1;
  return num < 0 ? 0 : num / s + 1;
}

template <int TT, int NT, int STRIDE, bool DW, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void tcs_kernel(const TcsArgs a) {
  constexpr int MT = TT / 32;     // 32-frame MFMA row tiles
  constexpr int M = TT / 16;      // 4-frame steps per lane run (4 runs per channel)
  constexpr int RUN = TT / 4;
  constexpr int NPP = (M - 1) * STRIDE + NKP;   // window pairs live per pass
  using Tile = DwTile<TT>;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const dwt = smem;                          // [KC][TT] bf16 swizzled
  char* const xs = smem + KC * Tile::ROWB;         // [KC][xpitch] bf16 (DW only)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * TT;
  const int cot0 = (blockIdx.z * 4 + wave) * NT;   // first 32-channel output tile of this wave
  const int n_cot = (a.c_out + 31) >> 5;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- pointwise accumulate of the 64-channel chunk currently in dwt --------------------------------
  auto pointwise = [&](const unsigned short* wfr, int kt_total, int chunk) {
    const int h = lane >> 5;
    const int g = (lane >> 4) & 1;
    const int q4 = (lane >> 2) & 3;
    const int p4 = lane & 3;
#pragma unroll
    for (int s = 0; s < KC / 16; ++s) {
      s16x8 bf[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int cot = cot0 + nt;
        if (cot < n_cot) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(
              wfr + ((size_t)(cot * kt_total + chunk * (KC / 16) + s) * 64 + lane) * 8);
          bf[nt] = __builtin_bit_cast(s16x8, v);
        } else {
          bf[nt] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
      s16x8 af[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int c = 16 * s + 8 * h + q4;
        const int t = 32 * mt + 16 * g + 4 * p4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (TS_LDS s16x4*)((TS_LDS char*)dwt + Tile::addr(c, t)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (TS_LDS s16x4*)((TS_LDS char*)dwt + Tile::addr(c + 4, t)));
        af[mt] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
    }
  };

  // ---- masked copy of a 64-channel chunk into dwt (identity "depthwise", stride rs) -----------------
  auto stage_identity = [&](const unsigned short* src, int c_total, int pitch, int len, int rs, int c0) {
    if (rs == 1) {
      constexpr int G = TT / 8;
      for (int idx = tid; idx < KC * G; idx += 256) {
        const int c = idx / G, gq = idx % G;
        const int t = t0 + gq * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (c0 + c < c_total && t < pitch) {
          v = *reinterpret_cast<const u32x4*>(src + ((size_t)(b * c_total + c0 + c) * pitch + t));
          v = keep_first(v, len - t);
        }
        *reinterpret_cast<u32x4*>(dwt + Tile::addr(c, gq * 8)) = v;
      }
    } else {
      for (int idx = tid; idx < KC * TT; idx += 256) {
        const int c = idx / TT, tl = idx % TT;
        const int ti = (t0 + tl) * rs;
        unsigned short v = 0;
        if (c0 + c < c_total && ti < len && ti < pitch) v = src[(size_t)(b * c_total + c0 + c) * pitch + ti];
        *reinterpret_cast<unsigned short*>(dwt + Tile::addr(c, tl)) = v;
      }
    }
  };

  // =================================== main source ====================================================
  const int len_in = a.len_in[b];
  if constexpr (DW) {
    const int len_mid = conv_len(len_in, a.kernel, STRIDE, a.padding, a.dilation);
    const int nk = a.npass * NKP;
    const int cw = wave * 16 + (lane >> 2);      // channel inside the chunk handled by this lane
    const int q = lane & 3;                      // time run (B column) and Toeplitz row (A row)
    const int G = a.xe >> 3;
    const int tin0 = t0 * STRIDE - a.padl8;
    for (int c0 = 0, chunk = 0; c0 < a.c_in; c0 += KC, ++chunk) {
      // ---- stage the input window of this chunk: xs[c][e] = mask(x)[c0 + c][tin0 + e]
      for (int idx = tid; idx < KC * G; idx += 256) {
        const int c = idx / G, gq = idx - c * G;
        const int t = tin0 + gq * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (c0 + c < a.c_in && t >= 0 && t < a.pitch_in) {
          v = *reinterpret_cast<const u32x4*>(a.x + ((size_t)(b * a.c_in + c0 + c) * a.pitch_in + t));
          v = keep_first(v, len_in - t);
        }
        u32x2* dst = reinterpret_cast<u32x2*>(xs + ((size_t)c * a.xpitch + gq * 8) * 2);
        dst[0] = u32x2{v[0], v[1]};
        dst[1] = u32x2{v[2], v[3]};
      }
      __syncthreads();
      // ---- depthwise on v_mfma_f32_4x4x4_16b_bf16
      {
        f32x4 d[M];
#pragma unroll
        for (int m = 0; m < M; ++m) d[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* xrow = xs + ((size_t)cw * a.xpitch + a.woff + q * RUN * STRIDE) * 2;
        const unsigned short* tp = a.taps + ((size_t)((c0 + cw) * 4 + q) * nk) * 4;
        for (int pass = 0; pass < a.npass; ++pass) {
          s16x4 A[NKP];
#pragma unroll
          for (int s = 0; s < NKP; ++s)
            A[s] = *reinterpret_cast<const s16x4*>(tp + (pass * NKP + s) * 4);
          s16x4 P[NPP];
#pragma unroll
          for (int u = 0; u < NPP; ++u)
            P[u] = *reinterpret_cast<const s16x4*>(xrow + (pass * NKP + u) * 8);
#pragma unroll
          for (int s = 0; s < NKP; ++s)
#pragma unroll
            for (int m = 0; m < M; ++m)
              d[m] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(A[s], P[m * STRIDE + s], d[m], 0, 0, 0);
        }
        // mask frames >= len_mid (quirk A2: the pointwise conv sees a re-masked input) and store
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const int tl = q * RUN + 4 * m;
          const int t = t0 + tl;
          f32x4 v = d[m];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = (t + i < len_mid) ? v[i] : 0.f;
          *reinterpret_cast<u32x2*>(dwt + Tile::addr(cw, tl)) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
        }
      }
      __syncthreads();
      pointwise(a.pw_w, a.kt_main, chunk);
    }
  } else {
    for (int c0 = 0, chunk = 0; c0 < a.c_in; c0 += KC, ++chunk) {
      if (chunk > 0) __syncthreads();
      stage_identity(a.x, a.c_in, a.pitch_in, len_in, STRIDE, c0);
      __syncthreads();
      pointwise(a.pw_w, a.kt_main, chunk);
    }
  }

  // =================================== residual source ================================================
  if (a.c_res > 0) {
    const int len_res = a.len_res[b];
    for (int c0 = 0, chunk = 0; c0 < a.c_res; c0 += KC, ++chunk) {
      __syncthreads();
      stage_identity(a.xres, a.c_res, a.pitch_res, len_res, a.res_stride, c0);
      __syncthreads();
      pointwise(a.res_w, a.kt_res, chunk);
    }
  }

  // =================================== epilogue =======================================================
  const int h = lane >> 5;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = (cot0 + nt) * 32 + (lane & 31);
    if (co < a.c_out) {
      const float bv = a.bias[co];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int t = t0 + 32 * mt + 8 * rg + 4 * h;
          float v0 = acc[mt][nt][4 * rg + 0] + bv, v1 = acc[mt][nt][4 * rg + 1] + bv;
          float v2 = acc[mt][nt][4 * rg + 2] + bv, v3 = acc[mt][nt][4 * rg + 3] + bv;
          if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
          if (t < a.pitch_out) {
            const size_t off = (size_t)(b * a.c_out + co) * a.pitch_out + t;
            if constexpr (OUT_F32) {
              *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + off) = f32x4{v0, v1, v2, v3};
            } else {
              *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(a.y) + off) =
                  u32x2{pack_bf16(v0, v1), pack_bf16(v2, v3)};
            }
          }
        }
      }
    }
  }
}

template <int TT, int NT, int STRIDE, bool DW, bool OUT_F32>
static int launch(const TcsArgs& a, int batch, hipStream_t stream) {
  constexpr int CO_WG = 4 * NT * 32;
  dim3 grid((a.t_out + TT - 1) / TT, batch, (round_up(a.c_out, 32) + CO_WG - 1) / CO_WG);
  size_t lds = (size_t)KC * TT * 2 + (DW ? (size_t)KC * a.xpitch * 2 : 0);
  auto kern = tcs_kernel<TT, NT, STRIDE, DW, OUT_F32>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
  return hip_status(hipGetLastError());
}

}  // namespace ts

extern "C" int ts_time_pitch(int T) { return ts::round_up(T < 1 ? 1 : T, 128); }

extern "C" int ts_tcs_subblock_fwd(const ts_tcs_desc* d, const void* x, const int32_t* len_in, const void* x_res,
                                   const int32_t* len_res, void* y, void* stream_) {
  using namespace ts;
  if (!d || !x || !y || !len_in || !d->pw_w || !d->bias) return TS_EINVAL;
  if (d->batch <= 0 || d->c_in <= 0 || d->c_out <= 0 || d->t_out <= 0) return TS_EINVAL;
  if (d->pitch_in % 8 || d->pitch_out % 8 || d->pitch_out < d->t_out) return TS_EINVAL;
  if (d->stride < 1 || d->dilation < 1 || d->kernel < 1) return TS_EINVAL;
  if (d->stride > 1 && d->dilation > 1) return TS_EINVAL;          // blocks.py:192-193
  if (d->c_res > 0 && (!x_res || !len_res || !d->res_w || d->pitch_res % 8)) return TS_EINVAL;
  if (!d->depthwise && d->kernel != 1) return TS_EUNSUPPORTED;     // dense K>1 convs are not on the hot path
  if (d->depthwise && (!d->dw_taps || d->dw_ksteps <= 0 || d->dw_ksteps % NKP)) return TS_EINVAL;
  if (d->depthwise && d->stride > 2) return TS_EUNSUPPORTED;
  if (d->out_fp32 && d->depthwise) return TS_EUNSUPPORTED;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);

  TcsArgs a{};
  a.x = static_cast<const unsigned short*>(x);
  a.xres = static_cast<const unsigned short*>(x_res);
  a.y = y;
  a.len_in = len_in;
  a.len_res = len_res;
  a.taps = static_cast<const unsigned short*>(d->dw_taps);
  a.pw_w = static_cast<const unsigned short*>(d->pw_w);
  a.res_w = static_cast<const unsigned short*>(d->res_w);
  a.bias = d->bias;
  a.c_in = d->c_in; a.c_out = d->c_out; a.c_res = d->c_res;
  a.pitch_in = d->pitch_in; a.pitch_out = d->pitch_out; a.pitch_res = d->pitch_res;
  a.t_out = d->t_out;
  a.kernel = d->kernel; a.stride = d->stride; a.dilation = d->dilation; a.padding = d->padding;
  a.relu = d->relu;
  a.res_stride = d->res_stride < 1 ? 1 : d->res_stride;
  a.kt_main = round_up(d->c_in, KC) / 16;
  a.kt_res = round_up(d->c_res > 0 ? d->c_res : 1, KC) / 16;

  const bool wide = round_up(d->c_out, 32) > 256;   // 512-channel tiles for the wide layers
  const int TT = wide ? 64 : 128;
  if (d->depthwise) {
    a.npass = d->dw_ksteps / NKP;
    const int padl4 = round_up(d->padding, 4);
    a.padl8 = round_up(padl4, 8);
    a.woff = a.padl8 - padl4;
    a.xe = round_up(TT * d->stride + 4 * d->dw_ksteps + 8, 8);
    const int xpd = round_up(a.xe / 2 - 2, 16) + 2;   // dwords per row == 2 (mod 16): conflict-free window reads
    a.xpitch = xpd * 2;
    if (d->stride == 1)
      return wide ? launch<64, 4, 1, true, false>(a, d->batch, stream) : launch<128, 2, 1, true, false>(a, d->batch, stream);
    return wide ? launch<64, 4, 2, true, false>(a, d->batch, stream) : launch<128, 2, 2, true, false>(a, d->batch, stream);
  }
  // pointwise only: `stride` is handled by the staging (generic gather when > 1)
  if (d->out_fp32) {
    if (d->stride != 1) return TS_EUNSUPPORTED;
    return launch<128, 2, 1, false, true>(a, d->batch, stream);
  }
  if (d->stride == 1)
    return wide ? launch<64, 4, 1, false, false>(a, d->batch, stream) : launch<128, 2, 1, false, false>(a, d->batch, stream);
  if (d->stride == 2)
    return wide ? launch<64, 4, 2, false, false>(a, d->batch, stream) : launch<128, 2, 2, false, false>(a, d->batch, stream);
  return TS_EUNSUPPORTED;
}
