// wav2vec2 encoder, first (unfused, fp32) version: what the reference gets from transformers.Wav2Vec2Model through
// _HuggingFaceEncoderAdapt.forward (huggingface/compatibility.py:31-42) for group-norm / post-LN checkpoints
// (facebook/wav2vec2-base-960h, -large-960h).  Activations are TIME-MAJOR fp32 [B][T][C]; the reference's final
// transpose(-1, -2) is a view on the host side.
//
//   ts_w2v_conv0_fwd      conv(1 -> C, k, s) + GroupNorm(C groups: per (clip, channel) over time) + GELU, three launches:
//                         partial sums (conv recomputed, never stored un-normalised), fp64 finalize, apply
//   ts_w2v_conv_fwd       strided conv C -> C as one GEMM per tap (rows s*t + j of the time-major input are a strided
//                         matrix) accumulating in place, + GELU
//   ts_w2v_linear_fwd     y = act(x W^T + b [+ res])
//   ts_w2v_layernorm_fwd  y = LN(x [+ res])
//   ts_w2v_posconv_fwd    y = x + gelu(grouped conv(x) + b): one batched GEMM per tap over a zero-padded copy
//   ts_w2v_groupconv_fwd  y = grouped conv(x) + b, same padding: a layer of data2vec-audio's stacked positional convs
//   ts_w2v_glu_fwd        y = x[:, :c] * sigmoid(x[:, c:]): the activation of the adapter layers behind the encoder (config.add_adapter)
//   ts_w2v_attention_fwd  softmax(q k^T * scale [keys >= len masked]) v per (clip, head)
// Every GEMM is this library's own: csrc/gemm_nt.hip (bf16 operands, token-major) and csrc/gemm_f32.hip (f32 mode, odd shapes).
// precision 0: fp32 GEMMs (tight parity with the fp32 reference).  precision 1: the GEMM operands are bf16 (MFMA rate),
// accumulation, residual stream, normalisations and softmax stay fp32; every producer writes the bf16 copy its consumer
// needs next to (or instead of) the fp32 result, so no separate cast pass exists.  Fused MFMA attention is the follow-up.
#include "ts_common.hpp"

namespace ts {

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, below fp32 GELU noise; a third of the instructions of ocml's erff:
// the conv0 and epilogue kernels are bound by exactly this)
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float r = 1.f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erf_as(x * 0.70710678118654752f)); }

// ---------------------------------------------------------------------------------------------------------------------
// conv0 + GroupNorm + GELU
// ---------------------------------------------------------------------------------------------------------------------
constexpr int C0_FR = 256;          // frames per workgroup
constexpr int C0_KMAX = 16;

struct Conv0Args {
  const float* wave;                // [B][n]
  const float* w;                   // [C][k]
  const float* gamma;
  const float* beta;
  float* partial;                   // [B][chunks][C][2]
  float* stats;                     // [B][C][2] = (scale, shift)
  float* y;                         // [B][T0][C] (may be null when only the bf16 copy is wanted)
  unsigned short* y16;              // optional bf16 copy
  long long n;
  int t0, c, k, s, chunks;
  int plain;                        // 1: y = conv + beta (bias), no normalisation, no activation (layer-norm family)
  float eps;
};

// One WAVE = 128 channels (a lane owns the pair 2l, 2l + 1 of its group: packed-f32 FMAs) x a run of frames; the 4 waves of a workgroup cover 512
// channels of the same C0_FR frames.  The signal samples are wave-uniform, so they are SCALAR loads (index built from blockIdx and the loop counter
// only) and enter the FMAs as SGPR operands: no LDS staging, no broadcast reads -- the first form of this kernel (samples in LDS, one lane per
// channel, 10 broadcast ds_reads per output) was LDS-issue bound at 505 us (statistics) + 869 us (apply) for C5.
typedef float c0v2 __attribute__((ext_vector_type(2)));
// gelu_erf on a channel pair: the polynomial and the products as packed-f32 operations (v_pk_fma_f32 / v_pk_mul_f32), only the two reciprocals and
// the two exponentials per pair stay scalar; the same arithmetic as gelu_erf element by element
__device__ __forceinline__ c0v2 gelu_erf2(c0v2 x) {
  const c0v2 z = x * 0.70710678118654752f;
  const c0v2 az = c0v2{fabsf(z[0]), fabsf(z[1])};
  const c0v2 d = __builtin_elementwise_fma(c0v2{0.3275911f, 0.3275911f}, az, c0v2{1.f, 1.f});
  const c0v2 t = c0v2{__frcp_rn(d[0]), __frcp_rn(d[1])};
  c0v2 p = __builtin_elementwise_fma(t, c0v2{1.061405429f, 1.061405429f}, c0v2{-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(t, p, c0v2{1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(t, p, c0v2{-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(t, p, c0v2{0.254829592f, 0.254829592f});
  p = p * t;
  const c0v2 m = -(az * az);
  const c0v2 e = c0v2{__expf(m[0]), __expf(m[1])};
  const c0v2 r = __builtin_elementwise_fma(-p, e, c0v2{1.f, 1.f});            // erf(|z|)
  const c0v2 er = c0v2{copysignf(r[0], z[0]), copysignf(r[1], z[1])};
  const c0v2 hx = x * 0.5f;
  return __builtin_elementwise_fma(hx, er, hx);
}
constexpr int C0_FB = 8;            // frames per unrolled block
typedef float __attribute__((address_space(4))) C0ConstF;

// KT / ST: kernel size and stride as compile-time constants (10 / 5: every published wav2vec2), 0 = read them from the arguments
template <bool APPLY, int KT, int ST>
__global__ __launch_bounds__(256) void w2v_conv0_kernel(const Conv0Args a) {
  const int kk = KT ? KT : a.k, ss = ST ? ST : a.s;
  constexpr int KU = KT ? KT : C0_KMAX;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int f0 = chunk * C0_FR;
  const int nf = a.t0 - f0 < C0_FR ? a.t0 - f0 : C0_FR;
  const float* __restrict__ x = a.wave + (size_t)b * a.n + (size_t)f0 * ss;
  for (int cg = wave * 128; cg < a.c; cg += 512) {
    const int c = cg + 2 * lane;
    const bool ok0 = c < a.c, ok1 = c + 1 < a.c;
    c0v2 w[KU];
#pragma unroll
    for (int j = 0; j < KU; ++j)
      w[j] = c0v2{(ok0 && j < kk) ? a.w[(size_t)c * kk + j] : 0.f, (ok1 && j < kk) ? a.w[(size_t)(c + 1) * kk + j] : 0.f};
    c0v2 scale = c0v2{1.f, 1.f}, shift = c0v2{0.f, 0.f}, s1 = c0v2{0.f, 0.f}, s2 = c0v2{0.f, 0.f};
    if constexpr (APPLY) {
      if (a.plain) {
        shift = c0v2{(ok0 && a.beta) ? a.beta[c] : 0.f, (ok1 && a.beta) ? a.beta[c + 1] : 0.f};
      } else {
        const float* st = a.stats + ((size_t)b * a.c + c) * 2;
        scale = c0v2{ok0 ? st[0] : 0.f, ok1 ? st[2] : 0.f};
        shift = c0v2{ok0 ? st[1] : 0.f, ok1 ? st[3] : 0.f};
      }
    }
    auto frame = [&](int f) {
      // wave-uniform address in the CONSTANT address space: hipcc then issues scalar loads (a plain global pointer stays on the vector path because
      // the stores below might alias it)
      const C0ConstF* xs = (const C0ConstF*)(unsigned long long)(x + (size_t)f * ss);
      c0v2 v = c0v2{0.f, 0.f};
#pragma unroll
      for (int j = 0; j < KU; ++j)
        if (KT || j < kk) { const float sj = xs[j]; v = __builtin_elementwise_fma(w[j], c0v2{sj, sj}, v); }
      if constexpr (APPLY) {
        v = __builtin_elementwise_fma(v, scale, shift);
        if (!a.plain) v = gelu_erf2(v);
        const size_t o = ((size_t)b * a.t0 + f0 + f) * a.c + c;
        if (a.y) { if (ok1) *reinterpret_cast<c0v2*>(a.y + o) = v; else if (ok0) a.y[o] = v[0]; }
        if (a.y16) {
          const unsigned pk = pack_bf16(v[0], v[1]);
          if (ok1) *reinterpret_cast<unsigned*>(a.y16 + o) = pk; else if (ok0) a.y16[o] = (unsigned short)(pk & 0xffffu);
        }
      } else {
        s1 += v;
        s2 = __builtin_elementwise_fma(v, v, s2);
      }
    };
    int f = 0;
    for (; f + C0_FB <= nf; f += C0_FB) {
#pragma unroll
      for (int u = 0; u < C0_FB; ++u) frame(f + u);
    }
    for (; f < nf; ++f) frame(f);
    if constexpr (!APPLY) {
      float* p = a.partial + (((size_t)b * a.chunks + chunk) * a.c + c) * 2;
      if (ok0) { p[0] = s1[0]; p[1] = s2[0]; }
      if (ok1) { p[2] = s1[1]; p[3] = s2[1]; }
    }
  }
}

// mean / biased variance over time per (clip, channel) in fp64 -> (scale, shift) of the affine normalisation.  Workgroup = 64 channels x 4 chunk
// quarters (one channel per lane, its partials strided over the 4 waves), combined through LDS: the chunk loop of the first form (one thread per
// channel walking all 250 chunks) took 63 us for 16 x 512 channels.
__global__ __launch_bounds__(256) void w2v_conv0_finalize_kernel(const Conv0Args a) {
  __shared__ double red[4][64][2];
  const int b = blockIdx.y, lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  double s1 = 0.0, s2 = 0.0;
  if (c < a.c)
    for (int ch = q; ch < a.chunks; ch += 4) {
      const float* p = a.partial + (((size_t)b * a.chunks + ch) * a.c + c) * 2;
      s1 += (double)p[0];
      s2 += (double)p[1];
    }
  red[q][lane][0] = s1;
  red[q][lane][1] = s2;
  __syncthreads();
  if (q || c >= a.c) return;
  s1 = red[0][lane][0] + red[1][lane][0] + red[2][lane][0] + red[3][lane][0];
  s2 = red[0][lane][1] + red[1][lane][1] + red[2][lane][1] + red[3][lane][1];
  const double mu = s1 / a.t0;
  double var = s2 / a.t0 - mu * mu;
  var = var < 0.0 ? 0.0 : var;
  const double rs = 1.0 / sqrt(var + (double)a.eps);
  const double g = a.gamma[c];
  a.stats[((size_t)b * a.c + c) * 2] = (float)(rs * g);
  a.stats[((size_t)b * a.c + c) * 2 + 1] = (float)((double)a.beta[c] - mu * rs * g);
}

// ---------------------------------------------------------------------------------------------------------------------
// elementwise epilogues
// ---------------------------------------------------------------------------------------------------------------------
// y[r][c] = act(y[r][c] + bias[c]) (+ res[r][c]); ld = row pitch of y and res; n % 4 == 0 path is vectorised
__global__ __launch_bounds__(256) void w2v_bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                           const float* __restrict__ res, long long rows, int n, long long ld,
                                                           long long ld_res, int act, unsigned short* __restrict__ y16) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = n >> 2;
  if (idx >= rows * n4) return;
  const long long r = idx / n4;
  const int c = (int)(idx - r * n4) * 4;
  float4 v = *reinterpret_cast<float4*>(y + r * ld + c);
  if (bias) {
    const float4 bb = *reinterpret_cast<const float4*>(bias + c);
    v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
  }
  if (act & 1) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
  if (res) {
    const float4 rr = *reinterpret_cast<const float4*>(res + r * ld_res + c);
    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
  }
  if (!(act & 2)) *reinterpret_cast<float4*>(y + r * ld + c) = v;      // act & 2: y is scratch, only the bf16 copy is wanted
  if (y16) *reinterpret_cast<uint2*>(y16 + r * n + c) = uint2{pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)};     // dense [rows][n]
}

// one wavefront per row: y = LN(x (+ xbias) (+ res)) * w + b.  NV float4 per lane hold the row (c <= 256 NV, c % 4 == 0):
// one read of the inputs, fp32 statistics in registers (mean, then centred sum of squares), one write of each output.
template <int NV>
__global__ __launch_bounds__(256) void w2v_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            const float* __restrict__ xbias, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y, long long rows, int c,
                                                            float eps, unsigned short* __restrict__ y16, int act) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + row * c;
  const float* rr = res ? res + row * c : nullptr;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < NV; ++it) {
    const int i = (it * 64 + lane) * 4;
    v[it] = float4{0.f, 0.f, 0.f, 0.f};
    if (i < c) {
      v[it] = *reinterpret_cast<const float4*>(xr + i);
      if (rr) { const float4 r4 = *reinterpret_cast<const float4*>(rr + i); v[it].x += r4.x; v[it].y += r4.y; v[it].z += r4.z; v[it].w += r4.w; }
      if (xbias) { const float4 b4 = *reinterpret_cast<const float4*>(xbias + i); v[it].x += b4.x; v[it].y += b4.y; v[it].z += b4.z; v[it].w += b4.w; }
      s += (v[it].x + v[it].y) + (v[it].z + v[it].w);
    }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mu = s / c;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < NV; ++it) {
    const int i = (it * 64 + lane) * 4;
    if (i < c) {
      const float d0 = v[it].x - mu, d1 = v[it].y - mu, d2 = v[it].z - mu, d3 = v[it].w - mu;
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rs = rsqrtf(q / c + eps);
#pragma unroll
  for (int it = 0; it < NV; ++it) {
    const int i = (it * 64 + lane) * 4;
    if (i < c) {
      const float4 w4 = *reinterpret_cast<const float4*>(w + i), b4 = *reinterpret_cast<const float4*>(b + i);
      float4 o4 = float4{(v[it].x - mu) * rs * w4.x + b4.x, (v[it].y - mu) * rs * w4.y + b4.y, (v[it].z - mu) * rs * w4.z + b4.z,
                         (v[it].w - mu) * rs * w4.w + b4.w};
      if (act) o4 = float4{gelu_erf(o4.x), gelu_erf(o4.y), gelu_erf(o4.z), gelu_erf(o4.w)};
      if (y) *reinterpret_cast<float4*>(y + row * c + i) = o4;
      if (y16) *reinterpret_cast<uint2*>(y16 + row * c + i) = uint2{pack_bf16(o4.x, o4.y), pack_bf16(o4.z, o4.w)};
    }
  }
}

// rows >= len[b] of a [B][T][C] tensor become 0 (hidden_states[~attention_mask] = 0)
__global__ __launch_bounds__(256) void w2v_mask_rows_kernel(float* __restrict__ x, const int* __restrict__ len, int t, int c) {
  const int b = blockIdx.y;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int n = len[b] < 0 ? 0 : (len[b] < t ? len[b] : t);
  const long long total = (long long)(t - n) * c;
  if (idx < total) x[((size_t)b * t + n) * c + idx] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
// positional conv: zero-padded copy in, bias + GELU + residual out
// ---------------------------------------------------------------------------------------------------------------------
// xp: [B][T + k][C] with k/2 zero rows before and k - k/2 after each clip
template <typename T>
__global__ __launch_bounds__(256) void w2v_pad_rows_kernel(const float* __restrict__ x, T* __restrict__ xp, int t, int c, int k) {
  const int b = blockIdx.y;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)(t + k) * c;
  if (idx >= total) return;
  const long long r = idx / c;
  const int col = (int)(idx - r * c);
  const long long src = r - k / 2;
  const float v = (src >= 0 && src < t) ? x[((size_t)b * t + src) * c + col] : 0.f;
  if constexpr (sizeof(T) == 2) xp[(size_t)b * total + idx] = (T)(pack_bf16(v, 0.f) & 0xffffu);
  else xp[(size_t)b * total + idx] = v;
}

// PLAIN: y = conv + bias (a layer of Data2VecAudioPositionalConvEmbedding: its LayerNorm + GELU follow in ts_w2v_layernorm_fwd); else the wav2vec2 /
// hubert embedding y = x + gelu(conv + bias)
template <bool PLAIN>
__global__ __launch_bounds__(256) void w2v_posconv_finish_kernel(const float* __restrict__ x, const float* __restrict__ yp,
                                                                 const float* __restrict__ bias, float* __restrict__ y, int t,
                                                                 int c, int k) {
  const int b = blockIdx.y;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)t * c) return;
  const int col = (int)(idx % c);
  // conv output frame r of clip b was accumulated at padded row b (T + k) + r
  const float v = yp[(size_t)b * (t + k) * c + idx] + bias[col];
  y[(size_t)b * t * c + idx] = PLAIN ? v : x[(size_t)b * t * c + idx] + gelu_erf(v);
}

// ---------------------------------------------------------------------------------------------------------------------
// attention softmax: one wavefront per (clip, head, query) row of scores [B][H][T][T], in place
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void w2v_softmax_kernel(float* __restrict__ s, const int* __restrict__ key_len, int heads, int t,
                                                          float scale, unsigned short* __restrict__ p16) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long long rows = (long long)gridDim.y * heads * t;
  (void)rows;
  const int b = blockIdx.y;
  if (row >= (long long)heads * t) return;
  float* p = s + ((size_t)b * heads * t + row) * t;
  const int n = key_len ? (key_len[b] < t ? (key_len[b] < 0 ? 0 : key_len[b]) : t) : t;
  // the reference adds finfo.min to the masked keys: with at least one valid key they get probability exactly 0; with
  // none (len = 0) every key is "equally masked" and the softmax is uniform over all T keys
  const int lim = n > 0 ? n : t;
  float m = -3.0e38f;
  for (int i = lane; i < lim; i += 64) m = fmaxf(m, p[i] * scale);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float z = 0.f;
  for (int i = lane; i < lim; i += 64) z += __expf(p[i] * scale - m);
  for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o);
  const float rz = 1.f / z;
  unsigned short* q = p16 ? p16 + ((size_t)b * heads * t + row) * t : nullptr;
  for (int i = lane; i < t; i += 64) {
    const float v = i < lim ? __expf(p[i] * scale - m) * rz : 0.f;
    if (q) q[i] = (unsigned short)(pack_bf16(v, 0.f) & 0xffffu);
    else p[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused attention (precision 1, head_dim 64): softmax(q k^T * scale) v without materialising the [T][T] scores.
//   workgroup = 128 queries of one (clip, head), 4 waves x 32 queries; K / V tiles of 64 keys staged in LDS for all 4 waves.
//   Per 32-key sub-tile and wave, v_mfma_f32_32x32x16_bf16 throughout:
//     S^T[key][query] = K Q^T   -- transposed on purpose: a lane then owns ONE query column and 16 of the 32 keys, so the
//                                  row maximum / sum of the online softmax are in-lane plus one exchange with lane ^ 32;
//     O^T[d][query] += V^T P^T  -- A = V^T read out of the [key][d] tile with ds_read_b64_tr_b16, B = P^T straight from
//                                  the accumulator registers of the first product: the K rows are loaded in the order
//                                  (bits 2 and 3 of the row index swapped) that makes accumulator register i of lane half h
//                                  hold key 16 (i / 8) + 8 h + i % 8, which is exactly the B-operand slot order.
//   Probabilities are rounded to bf16 for the second product (as in the unfused path), everything else is fp32.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int FA_KT = 64;          // keys per staged tile
constexpr int FA_PITCH = 144;      // bytes per staged row: 64 bf16 + 16 (rows 36 banks apart: conflict-free b128 / tr reads)
constexpr int FA_QW = 128;         // queries per workgroup

struct FaArgs {
  const unsigned short* qkv;       // [B][T][3C] bf16
  unsigned short* ctx;             // [B][T][C] bf16
  const int* key_len;
  int t, c;
  float scale_log2e;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void w2v_flash_attn_kernel(const FaArgs a) {
  __shared__ __attribute__((aligned(16))) char ks_[FA_KT * FA_PITCH];
  __shared__ __attribute__((aligned(16))) char vs_[FA_KT * FA_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, head = blockIdx.y;
  const int q0 = blockIdx.x * FA_QW + wave * 32;
  const size_t rowp = (size_t)3 * a.c;
  const unsigned short* base = a.qkv + (size_t)b * a.t * rowp + (size_t)head * 64;
  int lim = a.t;
  if (a.key_len) {
    const int n = a.key_len[b] < a.t ? a.key_len[b] : a.t;
    lim = n > 0 ? n : a.t;                       // no valid key: the reference's softmax degenerates to all keys
  }
  const int half = lane >> 5, n32 = lane & 31;
  // B operand of the first product: Q^T, lane = (query n32, k-half): 8 consecutive d per k-step
  s16x8 qf[4];
  {
    const int qrow = q0 + n32 < a.t ? q0 + n32 : a.t - 1;
    const uint4* qp = reinterpret_cast<const uint4*>(base + (size_t)qrow * rowp + 8 * half);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = __builtin_bit_cast(s16x8, qp[2 * ks]);
  }
  f32x16 o[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[mt][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int pm = (n32 & ~12) | ((n32 & 4) << 1) | ((n32 & 8) >> 1);          // K row order: bits 2 and 3 swapped
  const int q4 = (lane >> 2) & 3, gq = (lane >> 4) & 1, p4 = lane & 3;
  const int v_off = (8 * half + q4) * FA_PITCH + (16 * gq + 4 * p4) * 2;     // transposing read of the V tile

  // (A register-prefetched, double-buffered variant of this loop was measured SLOWER: 176 VGPRs halve the occupancy, and this
  // kernel is bound by the softmax VALU work -- exp2 runs at quarter rate -- which only other resident waves can hide.)
  for (int k0 = 0; k0 < lim; k0 += FA_KT) {
    __syncthreads();                                                          // the previous tile has been consumed
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int chunk = tid + 256 * rep, r = chunk >> 3, cc = chunk & 7;
      const int key = k0 + r < a.t ? k0 + r : a.t - 1;
      const unsigned short* src = base + (size_t)key * rowp + cc * 8;
      *reinterpret_cast<uint4*>(ks_ + r * FA_PITCH + cc * 16) = *reinterpret_cast<const uint4*>(src + a.c);
      *reinterpret_cast<uint4*>(vs_ + r * FA_PITCH + cc * 16) = *reinterpret_cast<const uint4*>(src + 2 * a.c);
    }
    __syncthreads();
    const bool full = k0 + FA_KT <= lim;                                      // no masked key in this tile (uniform)
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (k0 + sub * 32 >= lim) break;                                        // uniform: nothing but masked keys
      f32x16 s;
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = 0.f;
      const char* kr = ks_ + (sub * 32 + pm) * FA_PITCH + half * 16;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const s16x8*>(kr + ks * 32), qf[ks], s, 0, 0, 0);
      // accumulator register i <-> key k0 + 32 sub + 16 (i / 8) + 8 half + i % 8
      if (!full) {
        const int kbase = k0 + sub * 32 + 8 * half;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (kbase + 16 * (i >> 3) + (i & 7) >= lim) s[i] = -INFINITY;
      }
      float mx = s[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx * a.scale_log2e);                   // scale > 0: max commutes with it; finite (>= 1 valid key)
      // bare v_exp_f32 (exp2): the arguments are <= 0 and a flushed denormal is a zero weight -- the library form's range handling
      // (compare, scale, select around every exp) was a third of this VALU-bound loop's instructions
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float rs = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = __builtin_amdgcn_exp2f(fmaf(s[i], a.scale_log2e, -m_new)); rs += s[i]; }
      l_run = l_run * alpha + rs;
      m_run = m_new;
      if (__any(alpha != 1.f)) {                                              // after the first tiles the running maximum rarely moves
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[mt][i] *= alpha;
      }
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        const unsigned p01 = pack_bf16(s[8 * ks2 + 0], s[8 * ks2 + 1]), p23 = pack_bf16(s[8 * ks2 + 2], s[8 * ks2 + 3]);
        const unsigned p45 = pack_bf16(s[8 * ks2 + 4], s[8 * ks2 + 5]), p67 = pack_bf16(s[8 * ks2 + 6], s[8 * ks2 + 7]);
        const s16x8 pb = __builtin_bit_cast(s16x8, uint4{p01, p23, p45, p67});
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const char* va = vs_ + (sub * 32 + 16 * ks2) * FA_PITCH + v_off + 64 * mt;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)va));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)va + 4 * FA_PITCH));
          const s16x8 vf = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, o[mt], 0, 0, 0);
        }
      }
    }
  }
  const float l = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.f / l;
  const int query = q0 + n32;
  if (query < a.t) {
    unsigned short* dst = a.ctx + ((size_t)b * a.t + query) * a.c + (size_t)head * 64 + 4 * half;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g)       // accumulator registers 4g .. 4g+3 <-> d = 32 mt + 8 g + 4 half + 0..3
        *reinterpret_cast<uint2*>(dst + 32 * mt + 8 * g) =
            uint2{pack_bf16(o[mt][4 * g] * inv, o[mt][4 * g + 1] * inv), pack_bf16(o[mt][4 * g + 2] * inv, o[mt][4 * g + 3] * inv)};
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Positional conv as an implicit GEMM on the matrix cores (precision 1, 64 channels per group: wav2vec2-large).
//   workgroup = 128 frames x 64 output channels of one (clip, group); the 128 + k - 1 input rows it needs are staged ONCE in
//   LDS and tap j simply reads rows j .. j + 127 of that window (no im2col, no per-tap restaging);
//   waves 2 x 2: 64 frames x 32 channels each, v_mfma_f32_32x32x16_bf16, A = window rows (ds_read_b128), B = tap weights
//   [co][ci] straight from L2, prefetched one tap ahead.  Epilogue: + bias, GELU, + x (fp32 residual), coalesced along channels.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PC_TT = 128;         // frames per workgroup
constexpr int PC_PITCH = 144;      // bytes per staged row (64 bf16 + 16)

struct PcArgs {
  const unsigned short* xp;        // [B][T + k][C] bf16, k/2 zero rows in front of each clip
  const unsigned short* w;         // [k][G][64 co][64 ci] bf16
  const float* bias;
  const float* x;                  // [B][T][C] fp32 (residual)
  float* y;
  int t, c, k, groups;
  float* z;                        // training (ts_w2v_posconv_train): the conv result before bias and GELU, or NULL
  int plain;                       // training, data gradient: y = x + conv (no bias, no GELU)
};

__global__ __launch_bounds__(256) void w2v_posconv_mfma_kernel(const PcArgs a) {
  extern __shared__ __attribute__((aligned(16))) char win[];                  // [PC_TT + k - 1][PC_PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, g = blockIdx.y, t0 = blockIdx.x * PC_TT;
  const int rows = PC_TT + a.k - 1;
  const int prow = a.t + a.k;
  const unsigned short* src = a.xp + ((size_t)b * prow + t0) * a.c + (size_t)g * 64;
  for (int chunk = tid; chunk < rows * 8; chunk += 256) {
    const int r = chunk >> 3, cc = chunk & 7;
    uint4 v = uint4{0u, 0u, 0u, 0u};
    if (t0 + r < prow) v = *reinterpret_cast<const uint4*>(src + (size_t)r * a.c + cc * 8);
    *reinterpret_cast<uint4*>(win + r * PC_PITCH + cc * 16) = v;
  }
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;                                    // 64 frames x 32 channels per wave
  const int half = lane >> 5, n32 = lane & 31;
  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
  const char* arow = win + (wm * 64 + n32) * PC_PITCH + half * 16;            // A: row = frame, 8 consecutive ci per k-step
  const uint4* wp = reinterpret_cast<const uint4*>(a.w + ((size_t)g * 64 + wn * 32 + n32) * 64 + 8 * half);
  const size_t tap_stride = (size_t)a.groups * 64 * 64 / 8;                   // uint4 units
  uint4 bf[4], bn[4], bnn[4];                    // weights of tap j, j + 1, j + 2: an L2 round trip is longer than one tap's MFMAs
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) bf[ks] = wp[2 * ks];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) bn[ks] = wp[(a.k > 1 ? tap_stride : 0) + 2 * ks];
  for (int j = 0; j < a.k; ++j) {
    if (j + 2 < a.k) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bnn[ks] = wp[(size_t)(j + 2) * tap_stride + 2 * ks];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const s16x8 af = *reinterpret_cast<const s16x8*>(arow + (j + 32 * mt) * PC_PITCH + ks * 32);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(s16x8, bf[ks]), acc[mt], 0, 0, 0);
      }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { bf[ks] = bn[ks]; bn[ks] = bnn[ks]; }
  }
  const int co = g * 64 + wn * 32 + n32;
  const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = t0 + wm * 64 + 32 * mt + 8 * (i >> 2) + 4 * half + (i & 3);
      if (t < a.t) {
        const size_t o = ((size_t)b * a.t + t) * a.c + co;
        if (a.plain) a.y[o] = a.x[o] + acc[mt][i];
        else {
          if (a.z) a.z[o] = acc[mt][i];
          a.y[o] = a.x[o] + gelu_erf(acc[mt][i] + bv);
        }
      }
    }
}

static inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }

// f32-accumulating GEMM on the f32 matrix-core instruction, any operand layout (csrc/gemm_f32.hip): the f32 mode and the shapes without a kernel of
// their own
int gemm_f32(hipStream_t stream, bool in_bf16, const void* a, long long a_rs, long long a_cs, long long sa, long long ska, const void* b,
             long long b_rs, long long b_cs, long long sb, long long skb, void* c, long long ldc, long long sc, bool out_bf16, const float* bias,
             int M, int N, int K, int nkb, int batch, bool beta);

// row-major y[M][N] (ldc) = x[M][K] (lda) W[N][K]^T (ldw) + beta y, batched with element strides.  bf16 = x and W are bf16;
// y is f32 (or bf16 when y16 is set), accumulation f32 either way.
static int gemm_nt(hipStream_t stream, bool bf16, long long m, int n, int k, const void* x, long long lda, long long sx, const void* w,
                   long long ldw, long long sw, void* y, long long ldc, long long sy, float beta, int batch, bool y16 = false) {
  return gemm_f32(stream, bf16, x, lda, 1, sx, 0, w, 1, ldw, sw, 0, y, ldc, sy, y16, nullptr, (int)m, n, k, 1, batch, beta != 0.f);
}

// our own token-major bf16 GEMM with the fused epilogue (csrc/gemm_nt.hip): the default of the bf16 mode since round 3
int gemm_nt_bf16(hipStream_t stream, const void* x, long long lda, long long sx, const void* w, long long ldw, const float* bias,
                 const float* res, long long ld_res, float* y, long long ldc, void* y16, long long ld16, long long sy, long long M, int N, int K,
                 int gelu, int batch, const void* wf);
}  // namespace ts

using namespace ts;
#define TS_STREAM hipStream_t stream = reinterpret_cast<hipStream_t>(stream_); (void)hipGetLastError()

static inline int conv_frames(long long n, int k, int s) { return n < k ? 0 : (int)((n - k) / s + 1); }

extern "C" int64_t ts_w2v_conv0_workspace_bytes(int32_t batch, int64_t n_samples, int32_t c, int32_t kernel, int32_t stride) {
  if (batch <= 0 || c <= 0 || kernel <= 0 || stride <= 0 || n_samples < kernel) return TS_EINVAL;
  const int t0 = conv_frames(n_samples, kernel, stride);
  const int chunks = (t0 + C0_FR - 1) / C0_FR;
  return (int64_t)batch * chunks * c * 2 * sizeof(float) + (int64_t)batch * c * 2 * sizeof(float);
}

extern "C" int ts_w2v_conv0_fwd(const float* wave, int32_t batch, int64_t n_samples, const float* w, const float* gn_w,
                                const float* gn_b, int32_t c, int32_t kernel, int32_t stride, float eps, float* y, void* y_bf16,
                                void* workspace, void* stream_) {
  if (!wave || !w || (gn_w && !gn_b) || (!y && !y_bf16) || !workspace || batch <= 0 || c <= 0 || stride <= 0 || n_samples < kernel) return TS_EINVAL;
  if (kernel <= 0 || kernel > C0_KMAX) return TS_EUNSUPPORTED;
  TS_STREAM;
  Conv0Args a{};
  a.wave = wave; a.w = w; a.gamma = gn_w; a.beta = gn_b; a.y = y; a.y16 = static_cast<unsigned short*>(y_bf16);
  a.n = n_samples; a.c = c; a.k = kernel; a.s = stride; a.eps = eps;
  a.t0 = conv_frames(n_samples, kernel, stride);
  a.chunks = (a.t0 + C0_FR - 1) / C0_FR;
  a.partial = static_cast<float*>(workspace);
  a.stats = a.partial + (size_t)batch * a.chunks * c * 2;
  if ((c & 1) && (y_bf16 || y)) { /* odd channel counts: the last lane of a row stores a single element (handled in the kernel) */ }
  a.plain = gn_w ? 0 : 1;
  if (!a.plain) {
    if (kernel == 10 && stride == 5) hipLaunchKernelGGL((w2v_conv0_kernel<false, 10, 5>), dim3(a.chunks, batch), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((w2v_conv0_kernel<false, 0, 0>), dim3(a.chunks, batch), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(w2v_conv0_finalize_kernel, dim3((c + 63) / 64, batch), dim3(256), 0, stream, a);
  }
  if (kernel == 10 && stride == 5) hipLaunchKernelGGL((w2v_conv0_kernel<true, 10, 5>), dim3(a.chunks, batch), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((w2v_conv0_kernel<true, 0, 0>), dim3(a.chunks, batch), dim3(256), 0, stream, a);
  return hip_status(hipGetLastError());
}

extern "C" int ts_w2v_conv_fwd(const void* x, int32_t batch, int32_t t_in, int32_t c_in, const void* w_taps, const float* bias,
                               int32_t c_out, int32_t kernel, int32_t stride, int32_t act, int32_t precision, float* y, void* y_bf16,
                               const void* w_frag, void* stream_) {
  if (!x || !w_taps || !y || batch <= 0 || c_in <= 0 || c_out <= 0 || kernel <= 0 || stride <= 0 || t_in < kernel) return TS_EINVAL;
  if (c_out % 4 || precision < 0 || precision > 1 || act < 0 || act > 1) return TS_EUNSUPPORTED;
  TS_STREAM;
  const int t_out = conv_frames(t_in, kernel, stride);
  const size_t es = precision ? 2 : 4;
  if (precision) {
    // bf16 operands: OUR GEMM (csrc/gemm_nt.hip), ONE launch for all clips (grid.y = clip) over all taps, K = kernel * c_in.  Output frame t
    // reads input rows stride t .. stride t + kernel - 1, contiguous in the time-major layout, so the im2col matrix IS the input with
    // row pitch stride * c_in; its rows overlap when kernel > stride, which a kernel that only ever uses the pitch does not mind.  Bias +
    // GELU in the epilogue; with y_bf16 only the bf16 result is written (the next layer's operand), else the f32 one (a LayerNorm
    // follows).  A shape the kernel declines is an error here (TS_EUNSUPPORTED), not a reason to call a vendor library.
    return gemm_nt_bf16(stream, x, (long long)stride * c_in, (long long)t_in * c_in, w_taps, (long long)kernel * c_in, bias, nullptr, 0,
                        y_bf16 ? nullptr : y, c_out, y_bf16, c_out, (long long)t_out * c_out, t_out, c_out, kernel * c_in, act != 0, batch, w_frag);
  }
  // f32 mode (the reference's arithmetic): the f32 matrix-core GEMM, taps `stride` at a time (rows stride t .. stride t + stride - 1 of the
  // time-major input are one contiguous K = stride * c_in operand row)
  // Output frame t reads input rows stride*t .. stride*t + kernel - 1, which are CONTIGUOUS in the time-major layout: with a
  // row pitch of stride * c_in the first `stride` taps are one [t_out x stride*c_in] matrix, so taps go `stride` at a time
  // (k = 3, s = 2: taps {0, 1} in one GEMM with K = 2 c_in, tap 2 in a second one accumulating; k = 2, s = 2: one GEMM).
  for (int j = 0; j < kernel; j += stride) {
    const int nt = kernel - j < stride ? kernel - j : stride;
    if (int st = gemm_nt(stream, false, t_out, c_out, nt * c_in, static_cast<const char*>(x) + (size_t)j * c_in * es,
                         (long long)stride * c_in, (long long)t_in * c_in, static_cast<const char*>(w_taps) + (size_t)j * c_in * es,
                         (long long)kernel * c_in, 0, y, c_out, (long long)t_out * c_out, j ? 1.f : 0.f, batch))
      return st;
  }
  const long long rows = (long long)batch * t_out;
  // with a bf16 copy requested the f32 buffer is only the GEMM accumulator: it is not written back after the epilogue
  if (bias || act || y_bf16)
    hipLaunchKernelGGL(w2v_bias_act_kernel, dim3(nblk(rows * (c_out / 4))), dim3(256), 0, stream, y, bias, (const float*)nullptr, rows,
                       c_out, (long long)c_out, 0LL, act | (y_bf16 ? 2 : 0), static_cast<unsigned short*>(y_bf16));
  return hip_status(hipGetLastError());
}

extern "C" int ts_w2v_linear_fwd(const void* x, int64_t lda, const void* w, const float* bias, const float* res, int64_t ld_res,
                                 float* y, int64_t ldc, void* y_bf16, int64_t rows, int32_t n, int32_t k, int32_t act, int32_t precision,
                                 const void* w_frag, void* stream_) {
  if (!x || !w || !y || rows <= 0 || n <= 0 || k <= 0 || lda < k || ldc < n || (res && ld_res < n)) return TS_EINVAL;
  if (n % 4 || ldc % 4 || (res && ld_res % 4) || act < 0 || act > 3 || precision < 0 || precision > 1) return TS_EUNSUPPORTED;
  if ((act & 2) && !y_bf16) return TS_EINVAL;
  TS_STREAM;
  if (precision)
    // bf16 operands: OUR GEMM with bias / GELU / residual in its epilogue; the f32 result is skipped when only the bf16 copy is wanted.
    // A shape it declines is an error (TS_EUNSUPPORTED): no vendor library on the bf16 path.
    return gemm_nt_bf16(stream, x, lda, 0, w, k, bias, res, ld_res, (act & 2) ? nullptr : y, ldc, y_bf16, n, 0, rows, n, k, act & 1, 1, w_frag);
  // f32 mode (the reference's arithmetic): the f32 matrix-core GEMM + one epilogue pass
  // res == y: accumulate into the residual stream in place (beta = 1 inside the GEMM) -- no separate add, no second tensor to read
  const bool inplace = res && static_cast<const void*>(res) == static_cast<const void*>(y) && ld_res == ldc;
  if (int st = gemm_nt(stream, false, rows, n, k, x, lda, 0, w, k, 0, y, ldc, 0, inplace ? 1.f : 0.f, 1)) return st;
  const float* res_e = inplace ? nullptr : res;
  if (bias || res_e || act || y_bf16)
    hipLaunchKernelGGL(w2v_bias_act_kernel, dim3(nblk(rows * (n / 4))), dim3(256), 0, stream, y, bias, res_e, (long long)rows, n,
                       (long long)ldc, (long long)ld_res, act, static_cast<unsigned short*>(y_bf16));
  return hip_status(hipGetLastError());
}

extern "C" int ts_w2v_layernorm_fwd(const float* x, const float* res, const float* xbias, const float* w, const float* b, float eps,
                                    int64_t rows, int32_t c, int32_t act, float* y, void* y_bf16, void* stream_) {
  if (!x || !w || !b || (!y && !y_bf16) || rows <= 0 || c <= 0) return TS_EINVAL;
  if (c % 4 || c > 4096) return TS_EUNSUPPORTED;
  TS_STREAM;
  const dim3 grid((unsigned)((rows + 3) / 4));
  unsigned short* y16 = static_cast<unsigned short*>(y_bf16);
#define TS_LN(NV_) hipLaunchKernelGGL(w2v_layernorm_kernel<NV_>, grid, dim3(256), 0, stream, x, res, xbias, w, b, y, (long long)rows, c, eps, y16, act)
  if (c <= 512) TS_LN(2); else if (c <= 1024) TS_LN(4); else if (c <= 2048) TS_LN(8); else TS_LN(16);
#undef TS_LN
  return hip_status(hipGetLastError());
}

extern "C" int ts_w2v_mask_rows(float* x, int32_t batch, int32_t t, int32_t c, const int32_t* len, void* stream_) {
  if (!x || !len || batch <= 0 || t <= 0 || c <= 0) return TS_EINVAL;
  TS_STREAM;
  hipLaunchKernelGGL(w2v_mask_rows_kernel, dim3(nblk((long long)t * c), batch), dim3(256), 0, stream, x, len, t, c);
  return hip_status(hipGetLastError());
}

namespace ts {
// GLU over the channel halves of a row: y[r][j] = x[r][j] * sigmoid(x[r][c + j]) -- the activation of Wav2Vec2AdapterLayer (Conv1d to 2c channels, then
// nn.functional.glu over them); four channels per thread, f32 result and optional bf16 copy
__global__ __launch_bounds__(256) void w2v_glu_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned short* __restrict__ y16, long long rows, int c) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int q = c >> 2;
  if (idx >= rows * q) return;
  const long long r = idx / q;
  const int j = (int)(idx - r * q) * 4;
  const f32x4 a = *reinterpret_cast<const f32x4*>(x + r * 2 * c + j);
  const f32x4 g = *reinterpret_cast<const f32x4*>(x + r * 2 * c + c + j);
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = a[i] / (1.f + __expf(-g[i]));
  *reinterpret_cast<f32x4*>(y + r * c + j) = o;
  if (y16) *reinterpret_cast<u32x2*>(y16 + r * c + j) = u32x2{pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
}
}  // namespace ts

extern "C" int ts_w2v_glu_fwd(const float* x, int64_t rows, int32_t c, float* y, void* y_bf16, void* stream_) {
  if (!x || !y || rows <= 0 || c <= 0) return TS_EINVAL;
  if (c % 4 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) || (y_bf16 && (reinterpret_cast<uintptr_t>(y_bf16) & 7))) return TS_EUNSUPPORTED;
  using namespace ts;
  TS_STREAM;
  (void)hipGetLastError();
  hipLaunchKernelGGL(w2v_glu_kernel, dim3(nblk(rows * (c / 4))), dim3(256), 0, stream, x, y, static_cast<unsigned short*>(y_bf16), (long long)rows, c);
  return hip_status(hipGetLastError());
}

extern "C" int64_t ts_w2v_posconv_workspace_bytes(int32_t batch, int32_t t, int32_t c, int32_t kernel) {
  if (batch <= 0 || t <= 0 || c <= 0 || kernel <= 0) return TS_EINVAL;
  return (int64_t)2 * batch * (t + kernel) * c * sizeof(float);
}

namespace ts {
static int posconv_impl(const float* x, int32_t batch, int32_t t, int32_t c, const void* w_taps, const float* bias, int32_t kernel, int32_t groups,
                        int32_t precision, float* y, void* workspace, void* stream_, bool plain);
}

extern "C" int ts_w2v_posconv_fwd(const float* x, int32_t batch, int32_t t, int32_t c, const void* w_taps, const float* bias,
                                  int32_t kernel, int32_t groups, int32_t precision, float* y, void* y_bf16, void* workspace,
                                  void* stream_) {
  (void)y_bf16;
  return ts::posconv_impl(x, batch, t, c, w_taps, bias, kernel, groups, precision, y, workspace, stream_, false);
}

extern "C" int ts_w2v_groupconv_fwd(const float* x, int32_t batch, int32_t t, int32_t c, const void* w_taps, const float* bias,
                                    int32_t kernel, int32_t groups, int32_t precision, float* y, void* workspace, void* stream_) {
  return ts::posconv_impl(x, batch, t, c, w_taps, bias, kernel, groups, precision, y, workspace, stream_, true);
}

static int ts::posconv_impl(const float* x, int32_t batch, int32_t t, int32_t c, const void* w_taps, const float* bias, int32_t kernel,
                            int32_t groups, int32_t precision, float* y, void* workspace, void* stream_, bool plain) {
  if (!x || !w_taps || !bias || !y || !workspace || batch <= 0 || t <= 0 || c <= 0 || kernel <= 0 || groups <= 0 || c % groups) return TS_EINVAL;
  if (precision < 0 || precision > 1) return TS_EUNSUPPORTED;
  TS_STREAM;
  const int cg = c / groups;
  const long long prow = (long long)t + kernel;                       // padded rows per clip
  // workspace: yp f32 [B (t+k)][c] first, then the padded copy (f32 or bf16)
  float* yp = static_cast<float*>(workspace);
  char* xp = reinterpret_cast<char*>(yp + (size_t)batch * prow * c);
  const size_t es = precision ? 2 : 4;
  if (precision) hipLaunchKernelGGL(w2v_pad_rows_kernel<unsigned short>, dim3(nblk(prow * c), batch), dim3(256), 0, stream, x,
                                    reinterpret_cast<unsigned short*>(xp), t, c, kernel);
  else hipLaunchKernelGGL(w2v_pad_rows_kernel<float>, dim3(nblk(prow * c), batch), dim3(256), 0, stream, x, reinterpret_cast<float*>(xp), t, c, kernel);
  const size_t win_lds = (size_t)(PC_TT + kernel - 1) * PC_PITCH;
  if (!plain && precision && cg == 64 && win_lds <= 64 * 1024) {           // the fused kernel's epilogue is the wav2vec2 one
    PcArgs pa{};
    pa.xp = reinterpret_cast<const unsigned short*>(xp); pa.w = static_cast<const unsigned short*>(w_taps); pa.bias = bias; pa.x = x; pa.y = y;
    pa.t = t; pa.c = c; pa.k = kernel; pa.groups = groups;
    hipLaunchKernelGGL(w2v_posconv_mfma_kernel, dim3((t + PC_TT - 1) / PC_TT, groups, batch), dim3(256), win_lds, stream, pa);
    return hip_status(hipGetLastError());
  }
  // all clips at once: output row r (over the padded row space) = sum_j xp[r + j] W_j^T, per group; rows between clips are waste
  const long long m = (long long)batch * prow - kernel;
  for (int j = 0; j < kernel; ++j) {
    if (int st = gemm_nt(stream, precision != 0, m, cg, cg, xp + (size_t)j * c * es, c, cg,
                         static_cast<const char*>(w_taps) + (size_t)j * groups * cg * cg * es, cg, (long long)cg * cg, yp, c, cg,
                         j ? 1.f : 0.f, groups))
      return st;
  }
  if (plain) hipLaunchKernelGGL(w2v_posconv_finish_kernel<true>, dim3(nblk((long long)t * c), batch), dim3(256), 0, stream, x, yp, bias, y, t, c, kernel);
  else hipLaunchKernelGGL(w2v_posconv_finish_kernel<false>, dim3(nblk((long long)t * c), batch), dim3(256), 0, stream, x, yp, bias, y, t, c, kernel);
  return hip_status(hipGetLastError());
}

namespace ts {
// bf16 copy of x [B][t][c] f32 with `front` zero rows before and prow - front - t after each clip (prow rows per clip), one row of zeros behind the last clip
__global__ __launch_bounds__(256) void posconv_pad16_kernel(const float* __restrict__ x, unsigned short* __restrict__ xp, int t, int c, int prow, int front, long long total) {
  const long long idx = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (idx >= total) return;
  const long long row = idx / c;
  const int col = (int)(idx - row * c);
  const long long b = row / prow;
  const int r = (int)(row - b * prow) - front;
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (r >= 0 && r < t && b * prow < total / c - 1) v = *reinterpret_cast<const f32x4*>(x + ((size_t)b * t + r) * c + col);
  *reinterpret_cast<u32x2*>(xp + idx) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
}
}  // namespace ts

namespace ts {
// Weight gradient of the positional conv on the matrix cores (mixed-precision fine-tuning):
//   dW[j][g][o][i] = sum_b sum_t dz[b][t][64 g + o] xp[b][t + j][64 g + i],   xp = x behind kernel / 2 zero rows per clip (the forward's padded copy)
// -- for one (group, tap) a 64 x 64 product contracted over all B T frames; the f32 GEMM of round 5 needed 2.6 ms for the 2 048 of them.  Workgroup =
// (8 taps, group), four waves of two taps each; a stage is 128 frames of one clip: the dz tile [128][64] and the xp window [128 + 8][64] (shared by the
// eight taps) in LDS as bf16, both operands out of them with transposing reads (rows = the contraction index t), v_mfma_f32_32x32x16_bf16, 128 f32
// accumulators per lane; the next stage's rows travel global -> registers while the current one is multiplied.  256 workgroups, each owning its
// outputs: no partials, no atomics.
constexpr int PW_TT = 128, PW_TAPS = 8, PW_WIN = PW_TT + PW_TAPS;
struct PwArgs {
  const unsigned short* dz;        // [B][T][C] bf16
  const unsigned short* xp;        // [B][T + k][C] bf16
  float* dw;                       // [k][G][64][64]
  int batch, t, c, k, groups;
};
__global__ __launch_bounds__(256) void w2v_posconv_wgrad_kernel(const PwArgs a) {
  __shared__ __attribute__((aligned(16))) char dzs[PW_TT * PC_PITCH];
  __shared__ __attribute__((aligned(16))) char xps[PW_WIN * PC_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = blockIdx.x * PW_TAPS, g = blockIdx.y;
  const int prow = a.t + a.k, n_ch = (a.t + PW_TT - 1) / PW_TT, S = a.batch * n_ch;
  f32x16 acc[2][2][2];
#pragma unroll
  for (int tp = 0; tp < 2; ++tp)
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][mo][ni][r] = 0.f;
  // staging: 16-byte chunks; the dz tile is 128 x 8 chunks = 4 per thread, the xp window 136 x 8 = 1 088 chunks = 4.25 per thread
  uint4 rz[4], rx[5];
  auto fetch = [&](int s) {
    const int b = s / n_ch, t0 = (s % n_ch) * PW_TT;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int chunk = tid + 256 * q, r = chunk >> 3, cc = chunk & 7;
      rz[q] = (t0 + r < a.t) ? *reinterpret_cast<const uint4*>(a.dz + ((size_t)b * a.t + t0 + r) * a.c + (size_t)g * 64 + cc * 8) : uint4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int chunk = tid + 256 * q, r = chunk >> 3, cc = chunk & 7;
      const int row = t0 + j0 + r;                        // row of the clip's padded copy
      rx[q] = (chunk < PW_WIN * 8 && row < prow) ? *reinterpret_cast<const uint4*>(a.xp + ((size_t)b * prow + row) * a.c + (size_t)g * 64 + cc * 8) : uint4{0u, 0u, 0u, 0u};
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int chunk = tid + 256 * q; *reinterpret_cast<uint4*>(dzs + (chunk >> 3) * PC_PITCH + (chunk & 7) * 16) = rz[q]; }
#pragma unroll
    for (int q = 0; q < 5; ++q) { const int chunk = tid + 256 * q; if (chunk < PW_WIN * 8) *reinterpret_cast<uint4*>(xps + (chunk >> 3) * PC_PITCH + (chunk & 7) * 16) = rx[q]; }
  };
  const int half = lane >> 5, q4 = (lane >> 2) & 3, gq = (lane >> 4) & 1, p4 = lane & 3;
  const int tr_off = (8 * half + q4) * PC_PITCH + (16 * gq + 4 * p4) * 2;        // transposing read: rows = contraction index, columns = M / N index
  auto tr8 = [&](const char* p) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)p + 4 * PC_PITCH));
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  if (S > 0) fetch(0);
  for (int s = 0; s < S; ++s) {
    __syncthreads();                                       // the previous stage has been multiplied
    stage();
    __syncthreads();
    if (s + 1 < S) fetch(s + 1);
#pragma unroll
    for (int ks = 0; ks < PW_TT / 16; ++ks) {
      s16x8 af[2], bf[2][2];
#pragma unroll
      for (int mo = 0; mo < 2; ++mo) af[mo] = tr8(dzs + 16 * ks * PC_PITCH + tr_off + 64 * mo);
#pragma unroll
      for (int tp = 0; tp < 2; ++tp)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bf[tp][ni] = tr8(xps + (16 * ks + 2 * wave + tp) * PC_PITCH + tr_off + 64 * ni);
#pragma unroll
      for (int tp = 0; tp < 2; ++tp)
#pragma unroll
        for (int mo = 0; mo < 2; ++mo)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) acc[tp][mo][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mo], bf[tp][ni], acc[tp][mo][ni], 0, 0, 0);
    }
  }
  // accumulator register r of block (mo, ni): o = 32 mo + (r & 3) + 8 (r >> 2) + 4 half, i = 32 ni + lane % 32
#pragma unroll
  for (int tp = 0; tp < 2; ++tp) {
    const int j = j0 + 2 * wave + tp;
    if (j >= a.k) continue;
    float* const out = a.dw + ((size_t)j * a.groups + g) * 64 * 64;
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(size_t)(32 * mo + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + 32 * ni + (lane & 31)] = acc[tp][mo][ni][r];
  }
}
}  // namespace ts

extern "C" int64_t ts_w2v_posconv_wgrad_workspace(int32_t batch, int32_t t, int32_t c, int32_t kernel) {
  if (batch <= 0 || t <= 0 || c <= 0 || kernel <= 0) return TS_EINVAL;
  return (((int64_t)batch * (t + kernel) + 1) * c * 2 + 15) / 16 * 16 + ((int64_t)batch * t + 1) * c * 2;
}

/* dw[j][g][o][i] = sum over clips and frames of dz[..][64 g + o] * x[.. + j - kernel / 2][64 g + i]; see include/thunder_speech_amd.h */
extern "C" int ts_w2v_posconv_wgrad(const float* dz, const float* x, int32_t batch, int32_t t, int32_t c, int32_t kernel, int32_t groups, float* dw, void* workspace,
                                    void* stream_) {
  if (!dz || !x || !dw || !workspace || batch <= 0 || t <= 0 || c <= 0 || kernel <= 0 || groups <= 0 || c % groups) return TS_EINVAL;
  if (c / groups != 64 || c % 4 || (reinterpret_cast<uintptr_t>(workspace) & 15) || (reinterpret_cast<uintptr_t>(dz) & 15) || (reinterpret_cast<uintptr_t>(x) & 15))
    return TS_EUNSUPPORTED;
  TS_STREAM;
  const int prow = t + kernel;
  unsigned short* const xp = static_cast<unsigned short*>(workspace);
  unsigned short* const dz16 = reinterpret_cast<unsigned short*>(static_cast<char*>(workspace) + (((int64_t)batch * prow + 1) * c * 2 + 15) / 16 * 16);
  const long long total_x = ((long long)batch * prow + 1) * c, total_z = ((long long)batch * t + 1) * c;
  hipLaunchKernelGGL(posconv_pad16_kernel, dim3((unsigned)((total_x / 4 + 255) / 256)), dim3(256), 0, stream, x, xp, t, c, prow, kernel / 2, total_x);
  hipLaunchKernelGGL(posconv_pad16_kernel, dim3((unsigned)((total_z / 4 + 255) / 256)), dim3(256), 0, stream, dz, dz16, t, c, t, 0, total_z);
  PwArgs a{dz16, xp, dw, batch, t, c, kernel, groups};
  hipLaunchKernelGGL(w2v_posconv_wgrad_kernel, dim3((kernel + PW_TAPS - 1) / PW_TAPS, groups), dim3(256), 0, stream, a);
  return hip_status(hipGetLastError());
}

extern "C" int64_t ts_w2v_posconv_train_workspace(int32_t batch, int32_t t, int32_t c, int32_t kernel) {
  if (batch <= 0 || t <= 0 || c <= 0 || kernel <= 0) return TS_EINVAL;
  return ((int64_t)batch * (t + kernel) + 1) * c * 2;
}

/* Positional conv of mixed-precision fine-tuning on the matrix-core kernel; see include/thunder_speech_amd.h */
extern "C" int ts_w2v_posconv_train(const float* src, const float* res, int32_t batch, int32_t t, int32_t c, const void* w_taps_bf16, const float* bias, int32_t kernel,
                                    int32_t groups, int32_t backward, float* y, float* z, void* workspace, void* stream_) {
  if (!src || !res || !w_taps_bf16 || !y || !workspace || batch <= 0 || t <= 0 || c <= 0 || kernel <= 1 || groups <= 0 || c % groups) return TS_EINVAL;
  if (!backward && !bias) return TS_EINVAL;
  const size_t win_lds = (size_t)(PC_TT + kernel - 1) * PC_PITCH;
  if (c / groups != 64 || c % 4 || win_lds > 64 * 1024 || (reinterpret_cast<uintptr_t>(workspace) & 15) || (reinterpret_cast<uintptr_t>(src) & 15)) return TS_EUNSUPPORTED;
  TS_STREAM;
  const int prow = t + kernel;
  // forward: kernel / 2 zero rows in front (padding = kernel / 2).  Data gradient: out[s] = sum_j' dz[s + j' - (kernel - 1 - kernel / 2)] Wb[j'] with
  // Wb[j'] = W[kernel - 1 - j']^T -- the same product over a copy padded with kernel - 1 - kernel / 2 rows in front
  const int front = backward ? kernel - 1 - kernel / 2 : kernel / 2;
  unsigned short* const xp = static_cast<unsigned short*>(workspace);
  const long long total = ((long long)batch * prow + 1) * c;
  hipLaunchKernelGGL(posconv_pad16_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, stream, src, xp, t, c, prow, front, total);
  PcArgs pa{};
  pa.xp = xp; pa.w = static_cast<const unsigned short*>(w_taps_bf16); pa.bias = backward ? nullptr : bias; pa.x = res; pa.y = y;
  pa.t = t; pa.c = c; pa.k = kernel; pa.groups = groups; pa.z = backward ? nullptr : z; pa.plain = backward ? 1 : 0;
  hipLaunchKernelGGL(w2v_posconv_mfma_kernel, dim3((t + PC_TT - 1) / PC_TT, groups, batch), dim3(256), win_lds, stream, pa);
  return hip_status(hipGetLastError());
}

extern "C" int64_t ts_w2v_attention_workspace_bytes(int32_t batch, int32_t t, int32_t heads, int32_t precision) {
  if (batch <= 0 || t <= 0 || heads <= 0) return TS_EINVAL;
  return (int64_t)batch * heads * t * t * (sizeof(float) + (precision ? 2 : 0));
}

extern "C" int ts_w2v_attention_fwd(const void* qkv, int32_t batch, int32_t t, int32_t c, int32_t heads, const int32_t* key_len,
                                    int32_t precision, void* ctx, void* workspace, void* stream_) {
  if (!qkv || !ctx || !workspace || batch <= 0 || t <= 0 || c <= 0 || heads <= 0 || c % heads) return TS_EINVAL;
  if (precision < 0 || precision > 1) return TS_EUNSUPPORTED;
  TS_STREAM;
  const int hd = c / heads;
  const bool bf = precision != 0;
  if (bf && hd == 64 && c % 8 == 0) {
    FaArgs f{};
    f.qkv = static_cast<const unsigned short*>(qkv); f.ctx = static_cast<unsigned short*>(ctx); f.key_len = key_len;
    f.t = t; f.c = c; f.scale_log2e = 1.4426950408889634f / sqrtf((float)hd);
    hipLaunchKernelGGL(w2v_flash_attn_kernel, dim3((t + FA_QW - 1) / FA_QW, heads, batch), dim3(256), 0, stream, f);
    return hip_status(hipGetLastError());
  }
  const size_t es = bf ? 2 : 4;
  float* s = static_cast<float*>(workspace);
  unsigned short* p16 = bf ? reinterpret_cast<unsigned short*>(s + (size_t)batch * heads * t * t) : nullptr;
  for (int b = 0; b < batch; ++b) {
    const char* q = static_cast<const char*>(qkv) + (size_t)b * t * 3 * c * es;
    // scores[query][key] = q . k : batched over the heads (head h = columns [h hd, (h+1) hd) of each third of a qkv row)
    if (int st = gemm_nt(stream, bf, t, t, hd, q, 3LL * c, hd, q + (size_t)c * es, 3LL * c, hd, s + (size_t)b * heads * t * t, t,
                         (long long)t * t, 0.f, heads))
      return st;
  }
  hipLaunchKernelGGL(w2v_softmax_kernel, dim3((unsigned)(((long long)heads * t + 3) / 4), batch), dim3(256), 0, stream, s, key_len, heads, t,
                     1.f / sqrtf((float)hd), p16);
  for (int b = 0; b < batch; ++b) {
    const char* v = static_cast<const char*>(qkv) + ((size_t)b * t * 3 * c + 2 * c) * es;
    const void* p = bf ? static_cast<const void*>(p16 + (size_t)b * heads * t * t) : static_cast<const void*>(s + (size_t)b * heads * t * t);
    void* out = static_cast<char*>(ctx) + (size_t)b * t * c * es;
    // ctx[query][d] = sum_key p[query][key] v[key][d], batched over the heads (head h = columns [h hd, (h+1) hd) of a v / ctx row); bf16 in -> bf16 out
    if (int st = gemm_f32(stream, bf, p, t, 1, (long long)t * t, 0, v, 3LL * c, 1, hd, 0, out, c, hd, bf, nullptr, t, hd, t, 1, heads, false)) return st;
  }
  return hip_status(hipGetLastError());
}
