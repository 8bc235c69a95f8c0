// wav2vec2 fine-tuning: the backward kernels (and the few forward ones training needs on their own) of the transformer encoder.
//
// The reference fine-tunes HuggingFace CTC models through the same training_step as QuartzNet (module.py:102-127) with the conv feature
// extractor frozen (huggingface/compatibility.py:27-28 -> transformers' freeze_feature_encoder), i.e. autograd runs through
// feature_projection, the positional conv and the transformer layers of transformers.Wav2Vec2Model (third-party; forward restated in
// oracle/w2v.py).  Here every product of that backward pass is the f32 matrix-core GEMM (csrc/gemm_f32.hip: N / T forms, batched over
// (clip, head), the tap loop of the grouped positional conv as its outer contraction loop), and this file holds the rest:
//   LayerNorm backward (dx, and per-wave partial sums of d gamma / d beta)          ts_w2v_layernorm_bwd
//   column sums (bias gradients; the finish of d gamma / d beta)                    ts_w2v_colsum
//   GELU forward / backward on the erf form the inference kernels use              ts_w2v_gelu_fwd / _bwd
//   softmax forward (scaled, key-masked; P is saved) / backward                     ts_w2v_softmax_fwd / _bwd
//   row padding / un-padding of the time axis (positional conv)                     ts_w2v_pad_rows
//   train-time masking with the learned embedding (mask_time_prob)                  ts_w2v_mask_embed_fwd / _bwd
//   y = a + b                                                                      ts_w2v_add
// Activations are f32 [rows][c], contiguous (the reference's arithmetic); rows = clips x frames.
#include "ts_common.hpp"
#include "ts_philox.hpp"

namespace ts {

namespace {

__device__ __forceinline__ float erf_as(float x) {          // Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 (the inference kernels' erf)
  const float ax = fabsf(x);
  const float t = 1.f / (1.f + 0.3275911f * ax);
  const float y = 1.f - (((((1.061405429f * t - 1.453152027f) * t) + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t * __expf(-ax * ax);
  return x < 0.f ? -y : y;
}
__device__ __forceinline__ float gelu_f(float z) { return 0.5f * z * (1.f + erf_as(z * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float z) {
  return 0.5f * (1.f + erf_as(z * 0.70710678118654752f)) + z * 0.3989422804014327f * __expf(-0.5f * z * z);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// LayerNorm backward, one wave per row, the row in registers (NV float4 per lane: c <= 256 NV).  s = x (+ res); xhat = (s - mean) rstd;
// g = dy gamma; dx = rstd (g - mean(g) - xhat mean(g xhat)).  Every wave keeps the sums of dy xhat and dy over ITS rows in registers and
// writes them once: part[wave][0][c] (d gamma), part[wave][1][c] (d beta); ts_w2v_colsum adds the waves' partials.
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
                                                     const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ part, long long rows,
                                                     int c, float eps, float* __restrict__ zero_a = nullptr, float* __restrict__ zero_b = nullptr) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * 4;
  if (zero_a && blockIdx.x == 0)                     // ts_w2v_layernorm_bwd_set: the reducing launch behind this one adds with atomics -- its destinations are zeroed here
    for (int i = threadIdx.x; i < c; i += 256) { zero_a[i] = 0.f; zero_b[i] = 0.f; }
  f32x4 dg[NV], db[NV], gm[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    dg[j] = db[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int col = 4 * (lane + 64 * j);
    gm[j] = col < c ? *reinterpret_cast<const f32x4*>(gamma + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (long long r = wave; r < rows; r += n_waves) {
    f32x4 s[NV], g[NV];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int col = 4 * (lane + 64 * j);
      s[j] = g[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (col < c) {
        s[j] = *reinterpret_cast<const f32x4*>(x + r * c + col);
        if (res) s[j] += *reinterpret_cast<const f32x4*>(res + r * c + col);
        g[j] = *reinterpret_cast<const f32x4*>(dy + r * c + col);
      }
      sum += s[j][0] + s[j][1] + s[j][2] + s[j][3];
    }
    const float mean = wave_sum(sum) / c;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int col = 4 * (lane + 64 * j);
      if (col < c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { s[j][i] -= mean; var += s[j][i] * s[j][i]; }
      }
    }
    const float rstd = rsqrtf(wave_sum(var) / c + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[j][i] *= rstd;                               // xhat
        dg[j][i] += g[j][i] * s[j][i];
        db[j][i] += g[j][i];
        g[j][i] *= gm[j][i];
        m1 += g[j][i];
        m2 += g[j][i] * s[j][i];
      }
    }
    m1 = wave_sum(m1) / c;
    m2 = wave_sum(m2) / c;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int col = 4 * (lane + 64 * j);
      if (col < c) {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = rstd * (g[j][i] - m1 - s[j][i] * m2);
        *reinterpret_cast<f32x4*>(dx + r * c + col) = o;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int col = 4 * (lane + 64 * j);
    if (col < c) {
      *reinterpret_cast<f32x4*>(part + (wave * 2 + 0) * c + col) = dg[j];
      *reinterpret_cast<f32x4*>(part + (wave * 2 + 1) * c + col) = db[j];
    }
  }
}

// out[j] (+)= sum_r x[r][j]: 256 threads = 64 column lanes x 4 row groups per workgroup; grid.y strides the rows; atomics add the groups' sums
// into the zeroed (or accumulating) output.  Small (bias-sized) outputs: the atomic traffic is c per workgroup.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long long rows, int c, long long ld) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  float s = 0.f;
  if (col < c)
    for (long long r = (long long)blockIdx.y * 4 + rg; r < rows; r += (long long)gridDim.y * 4) s += x[r * ld + col];
  __shared__ float sm[4][64];
  sm[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && col < c) atomicAdd(out + col, sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// y = gelu(z + bias[col]) (dy == NULL) or dz = dy gelu'(z + bias[col]); bias may be NULL; c % 4 == 0 with a bias
__global__ __launch_bounds__(256) void gelu_kernel(const float* __restrict__ z, const float* __restrict__ bias, int c, const float* __restrict__ dy,
                                                   float* __restrict__ out, long long n) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n) {
    f32x4 v = *reinterpret_cast<const f32x4*>(z + i);
    if (bias) v += *reinterpret_cast<const f32x4*>(bias + (i % c));
    f32x4 o;
    if (dy) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(dy + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = g[k] * gelu_df(v[k]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = gelu_f(v[k]);
    }
    *reinterpret_cast<f32x4*>(out + i) = o;
  } else {
    for (long long k = i; k < n; ++k) {
      const float v = z[k] + (bias ? bias[k % c] : 0.f);
      out[k] = dy ? dy[k] * gelu_df(v) : gelu_f(v);
    }
  }
}

// In-place softmax over the keys of one (clip, head, query) row per wave: p = softmax(scale s), keys >= key_len[clip] get probability 0
// (transformers adds finfo.min to them: the same after the softmax unless every key is padded, which a clip with at least one frame rules out).
// Rows of `pitch` >= t floats (a pitch that is a multiple of 4 keeps the attention GEMMs on their vector loads); columns t .. pitch are zeroed.
__global__ __launch_bounds__(256) void softmax_fwd_kernel(float* __restrict__ s, const int* __restrict__ key_len, int heads, int t, int pitch, float scale) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  if (row >= (long long)heads * t) return;
  float* p = s + ((long long)b * heads * t + row) * pitch;
  const int nk = key_len ? min(key_len[b], t) : t;
  float mx = -3.0e38f;
  for (int j = lane; j < nk; j += 64) mx = fmaxf(mx, p[j] * scale);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int j = lane; j < nk; j += 64) sum += __expf(p[j] * scale - mx);
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int j = lane; j < pitch; j += 64) p[j] = j < nk ? __expf(p[j] * scale - mx) * inv : 0.f;
}

// ds = scale p (dp - sum_k dp_k p_k), in place over dp; one wave per row
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp, long long rows, int t, int pitch, float scale) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* pr = p + row * pitch;
  float* dr = dp + row * pitch;
  float dot = 0.f;
  for (int j = lane; j < t; j += 64) dot += pr[j] * dr[j];
  dot = wave_sum(dot);
  for (int j = lane; j < pitch; j += 64) dr[j] = j < t ? scale * pr[j] * (dr[j] - dot) : 0.f;
}

// dst[b][r + left][:] = src[b][r][:] for r < t, every other row of dst (t_dst rows per clip) = 0; with `extract` the other way round:
// dst[b][r][:] (t rows) = src[b][r + left][:] (src has t_dst rows per clip)
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int t, int t_dst, int left, int c, int extract) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  const int b = blockIdx.y;
  const long long n = (long long)(extract ? t : t_dst) * c;
  if (i >= n) return;
  const int r = (int)(i / c), col = (int)(i % c);
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (extract) {
    v = *reinterpret_cast<const f32x4*>(src + ((long long)b * t_dst + r + left) * c + col);
    *reinterpret_cast<f32x4*>(dst + ((long long)b * t + r) * c + col) = v;
  } else {
    if (r >= left && r < left + t) v = *reinterpret_cast<const f32x4*>(src + ((long long)b * t + r - left) * c + col);
    *reinterpret_cast<f32x4*>(dst + ((long long)b * t_dst + r) * c + col) = v;
  }
}

// forward: rows with mask != 0 <- embed.  backward: dx = dy with those rows zeroed (in place over dy is fine), dembed += sum of dy over them.
__global__ __launch_bounds__(256) void mask_embed_kernel(float* __restrict__ x, const unsigned char* __restrict__ mask, const float* __restrict__ embed,
                                                         float* __restrict__ dembed, long long rows, int c) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= rows * c) return;
  const long long r = i / c;
  const int col = (int)(i % c);
  if (!mask[r]) return;
  float* px = x + i;
  if (dembed) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { atomicAdd(dembed + col + k, px[k]); px[k] = 0.f; }
  } else {
    *reinterpret_cast<f32x4*>(px) = *reinterpret_cast<const f32x4*>(embed + col);
  }
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long long n) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n) *reinterpret_cast<f32x4*>(y + i) = *reinterpret_cast<const f32x4*>(a + i) + *reinterpret_cast<const f32x4*>(b + i);
  else for (long long k = i; k < n; ++k) y[k] = a[k] + b[k];
}

inline unsigned blk4(long long n) { return (unsigned)(((n + 3) / 4 + 255) / 256); }
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace ts

using namespace ts;
#define TS_STREAM hipStream_t stream = reinterpret_cast<hipStream_t>(stream_); (void)hipGetLastError()

namespace ts {
// f32 rows -> bf16 copy and / or TRANSPOSED bf16 copy, the operands of the mixed-precision fine-tuning products (huggingface/train.py LinearMixed):
//   y[r][j] = bf16(x[r][j]);   yt[j][r] = bf16(x[r][j]) for r < rows, 0 for rows <= r < rows_pad
// (the transposed copy's contraction index -- the rows -- is padded with zeros to a multiple of 32, what ts_gemm_nt_bf16 wants).  64 x 64 tiles
// through LDS: 256-byte row reads, 128-byte writes in both layouts.
// colsum (may be NULL): colsum[j] += sum_r x[r][j] on the way (the bias gradient of a linear layer is the column sum of the very dy this kernel casts:
// one pass over dy instead of two, and no launch of its own -- 97 launches + 97 zero fills of a wav2vec2-large fine-tuning step)
__global__ __launch_bounds__(256) void cast_bf16_t_kernel(const float* __restrict__ x, long long ldx, int rows, int c, unsigned short* __restrict__ y,
                                                          long long ldy, unsigned short* __restrict__ yt, long long ldt, int rows_pad, float* __restrict__ colsum) {
  __shared__ unsigned short tile[64][66];
  __shared__ float csum[4][64];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  float acc = 0.f;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, cc = c0 + tx;
    unsigned short v = 0;
    if (r < rows && cc < c) {
      const float xv = x[(size_t)r * ldx + cc];
      acc += xv;
      v = (unsigned short)(pack_bf16(xv, 0.f) & 0xffffu);
      if (y) y[(size_t)r * ldy + cc] = v;
    }
    tile[i][tx] = v;
  }
  if (colsum) csum[ty][tx] = acc;
  __syncthreads();
  if (colsum && ty == 0 && c0 + tx < c && r0 < rows) atomicAdd(colsum + c0 + tx, (csum[0][tx] + csum[1][tx]) + (csum[2][tx] + csum[3][tx]));
  if (yt) {
    for (int i = ty; i < 64; i += 4) {
      const int cc = c0 + i, r = r0 + tx;
      if (cc < c && r < rows_pad) yt[(size_t)cc * ldt + r] = tile[tx][i];
    }
  }
}
}  // namespace ts

namespace ts {
__global__ __launch_bounds__(256) void w2v_sum_parts_kernel(const float* __restrict__ parts, float* __restrict__ out, long long n4, int n_parts, long long stride4,
                                                            const float* __restrict__ bias, int c4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 acc = reinterpret_cast<const f32x4*>(parts)[i];
  for (int p = 1; p < n_parts; ++p) acc += reinterpret_cast<const f32x4*>(parts)[(long long)p * stride4 + i];      // fixed order: deterministic
  if (bias) acc += reinterpret_cast<const f32x4*>(bias)[i % c4];
  reinterpret_cast<f32x4*>(out)[i] = acc;
}
}  // namespace ts

/* out[r][j] = sum_p parts[p][r][j] + bias[j]  (rows of c floats, c % 4 == 0): the split-K form of a linear layer's forward product */
extern "C" int ts_w2v_sum_parts_bias(const float* parts, const float* bias, int32_t c, float* out, int64_t n, int32_t n_parts, void* stream_) {
  if (!parts || !bias || !out || n <= 0 || c <= 0 || c % 4 || n % c || n_parts < 1 || (reinterpret_cast<uintptr_t>(parts) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) ||
      (reinterpret_cast<uintptr_t>(bias) & 15))
    return TS_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::w2v_sum_parts_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), parts, out, (long long)(n / 4), (int)n_parts,
                     (long long)(n / 4), bias, (int)(c / 4));
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_w2v_sum_parts(const float* parts, float* out, int64_t n, int32_t n_parts, void* stream_) {
  if (!parts || !out || n <= 0 || n % 4 || n_parts < 1 || (reinterpret_cast<uintptr_t>(parts) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return TS_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::w2v_sum_parts_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), parts, out, (long long)(n / 4), (int)n_parts,
                     (long long)(n / 4), (const float*)nullptr, 1);
  return ts::hip_status(hipGetLastError());
}

namespace ts {
// The activation between the two feed-forward linears of mixed-precision fine-tuning, fused with the operand cast of the second one:
//   a[r][j] = dropout(gelu(z[r][j] + bias[j]))  ->  y16[r][j] = bf16(a), yt16[j][r] = bf16(a)      (rows >= `rows` of the transposed copy: zeros)
// -- the f32 activation (64 MB per layer at 8 x 10 s x 4096) has no other reader: gelu (read + write), dropout (read + write) and cast (read + 2 half writes)
// were 448 MB per layer, this is 128.  Dropout: ts_train_dropout's mask (element e = r c + j: word e & 3 of Philox block e >> 2, keep iff u01 >= p); a
// thread owns four consecutive columns = one block.  64 x 64 tiles through LDS for the transposed copy (as cast_bf16_t_kernel).
__global__ __launch_bounds__(256) void ffn_act_cast_kernel(const float* __restrict__ z, const float* __restrict__ bias, int rows, int c, float p, float scale,
                                                           unsigned long long seed, unsigned short* __restrict__ y, unsigned short* __restrict__ yt, long long ldt,
                                                           int rows_pad) {
  __shared__ unsigned short tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx4 = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int i = ty + 16 * pass, r = r0 + i, cc = c0 + 4 * tx4;
    unsigned lo = 0, hi = 0;
    if (r < rows && cc < c) {
      f32x4 v = *reinterpret_cast<const f32x4*>(z + (size_t)r * c + cc);
      if (bias) v += *reinterpret_cast<const f32x4*>(bias + cc);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = gelu_f(v[k]);
      if (p > 0.f) {
        const Philox4 rn = philox(seed, PHILOX_DROPOUT, ((unsigned long long)r * c + cc) >> 2);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = u01(rn.v[k]) >= p ? v[k] * scale : 0.f;
      }
      lo = pack_bf16(v[0], v[1]); hi = pack_bf16(v[2], v[3]);
      if (y) *reinterpret_cast<u32x2*>(y + (size_t)r * c + cc) = u32x2{lo, hi};
    }
    tile[i][4 * tx4 + 0] = (unsigned short)(lo & 0xffffu); tile[i][4 * tx4 + 1] = (unsigned short)(lo >> 16);
    tile[i][4 * tx4 + 2] = (unsigned short)(hi & 0xffffu); tile[i][4 * tx4 + 3] = (unsigned short)(hi >> 16);
  }
  __syncthreads();
  if (yt) {                                          // 4-byte stores of two consecutive rows (rows_pad and ldt are even)
    const int tp = threadIdx.x & 31, tg = threadIdx.x >> 5;
    for (int i = tg; i < 64; i += 8) {
      const int cc = c0 + i, r = r0 + 2 * tp;
      if (cc < c && r + 1 < rows_pad) *reinterpret_cast<unsigned*>(yt + (size_t)cc * ldt + r) = (unsigned)tile[2 * tp][i] | ((unsigned)tile[2 * tp + 1][i] << 16);
      else if (cc < c && r < rows_pad) yt[(size_t)cc * ldt + r] = tile[2 * tp][i];
    }
  }
}

// backward of the same: dz = dropout_bwd(da) * gelu'(z + bias)  (the mask re-drawn from the seed)
__global__ __launch_bounds__(256) void ffn_act_bwd_kernel(const float* __restrict__ z, const float* __restrict__ bias, int c, const float* __restrict__ da, float p, float scale,
                                                          unsigned long long seed, float* __restrict__ dz, long long n4) {
  const long long i4 = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i4 >= n4) return;
  const long long e = i4 * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(z + e);
  if (bias) v += *reinterpret_cast<const f32x4*>(bias + (e % c));
  f32x4 g = *reinterpret_cast<const f32x4*>(da + e);
  if (p > 0.f) {
    const Philox4 rn = philox(seed, PHILOX_DROPOUT, (unsigned long long)i4);
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = u01(rn.v[k]) >= p ? g[k] * scale : 0.f;
  }
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = g[k] * gelu_df(v[k]);
  *reinterpret_cast<f32x4*>(dz + e) = o;
}
}  // namespace ts

/* y16 / yt16 = bf16(dropout(gelu(z + bias))) as ts_w2v_cast_bf16_t would cast it; see include/thunder_speech_amd.h */
extern "C" int ts_w2v_ffn_act_cast(const float* z, const float* bias, int64_t rows, int32_t c, float p_drop, uint64_t seed, void* y, void* yt, int64_t ldt, int64_t rows_pad,
                                   void* stream_) {
  if (!z || (!y && !yt) || rows <= 0 || c <= 0 || rows >= (1ll << 31) || !(p_drop >= 0.f && p_drop < 1.f)) return TS_EINVAL;
  if (yt && (rows_pad < rows || ldt < rows_pad || rows_pad >= (1ll << 31))) return TS_EINVAL;
  if (c % 4 || !al16(z) || (bias && !al16(bias)) || (y && (reinterpret_cast<uintptr_t>(y) & 7)) || (yt && (ldt % 2 || (reinterpret_cast<uintptr_t>(yt) & 3)))) return TS_EUNSUPPORTED;
  TS_STREAM;
  const long long rp = yt ? rows_pad : rows;
  hipLaunchKernelGGL(ts::ffn_act_cast_kernel, dim3((unsigned)((c + 63) / 64), (unsigned)((rp + 63) / 64)), dim3(256), 0, stream, z, bias, (int)rows, (int)c, p_drop,
                     1.f / (1.f - p_drop), (unsigned long long)seed, static_cast<unsigned short*>(y), static_cast<unsigned short*>(yt), (long long)ldt, (int)rows_pad);
  return ts::hip_status(hipGetLastError());
}

/* dz = dropout_backward(da) * gelu'(z + bias), n = rows * c elements; see include/thunder_speech_amd.h */
extern "C" int ts_w2v_ffn_act_bwd(const float* z, const float* bias, int32_t c, const float* da, float p_drop, uint64_t seed, float* dz, int64_t n, void* stream_) {
  if (!z || !da || !dz || n <= 0 || c <= 0 || n % c || !(p_drop >= 0.f && p_drop < 1.f)) return TS_EINVAL;
  if (c % 4 || !al16(z) || !al16(da) || !al16(dz) || (bias && !al16(bias))) return TS_EUNSUPPORTED;
  TS_STREAM;
  hipLaunchKernelGGL(ts::ffn_act_bwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, z, bias, c, da, p_drop, 1.f / (1.f - p_drop), (unsigned long long)seed, dz,
                     (long long)(n / 4));
  return ts::hip_status(hipGetLastError());
}

namespace ts {
// cast_bf16_t_kernel with 16-byte loads and 8-byte / 4-byte stores (c % 4 == 0, 16-byte aligned rows): a thread owns four consecutive columns of four rows of the
// 64 x 64 tile; the transposed copy leaves as 4-byte stores of two consecutive rows.  The scalar form above moved 32 MB in 14.9 us (2.1 TB/s): 315 launches = 4.7 ms
// of a 31-ms fine-tuning step.
__global__ __launch_bounds__(256) void cast_bf16_t4_kernel(const float* __restrict__ x, long long ldx, int rows, int c, unsigned short* __restrict__ y, long long ldy,
                                                           unsigned short* __restrict__ yt, long long ldt, int rows_pad, float* __restrict__ colsum) {
  __shared__ unsigned short tile[64][66];
  __shared__ float csum[16][64];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx4 = threadIdx.x & 15, ty = threadIdx.x >> 4;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int i = ty + 16 * pass, r = r0 + i, cc = c0 + 4 * tx4;
    unsigned lo = 0, hi = 0;
    if (r < rows && cc < c) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * ldx + cc);
      acc += v;
      lo = pack_bf16(v[0], v[1]); hi = pack_bf16(v[2], v[3]);
      if (y) *reinterpret_cast<u32x2*>(y + (size_t)r * ldy + cc) = u32x2{lo, hi};
    }
    tile[i][4 * tx4 + 0] = (unsigned short)(lo & 0xffffu); tile[i][4 * tx4 + 1] = (unsigned short)(lo >> 16);
    tile[i][4 * tx4 + 2] = (unsigned short)(hi & 0xffffu); tile[i][4 * tx4 + 3] = (unsigned short)(hi >> 16);
  }
  if (colsum) {
#pragma unroll
    for (int k = 0; k < 4; ++k) csum[ty][4 * tx4 + k] = acc[k];
  }
  __syncthreads();
  if (colsum && threadIdx.x < 64 && c0 + (int)threadIdx.x < c && r0 < rows) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += csum[q][threadIdx.x];
    atomicAdd(colsum + c0 + threadIdx.x, t);
  }
  if (yt) {
    // lane pair-rows: thread (tp = tid & 31 -> rows 2 tp, 2 tp + 1; tg = tid >> 5 -> columns tg, tg + 8, ...)
    const int tp = threadIdx.x & 31, tg = threadIdx.x >> 5;
    for (int i = tg; i < 64; i += 8) {
      const int cc = c0 + i, r = r0 + 2 * tp;
      if (cc < c && r < rows_pad) {
        const unsigned v = (unsigned)tile[2 * tp][i] | ((unsigned)tile[2 * tp + 1][i] << 16);
        if (r + 1 < rows_pad) *reinterpret_cast<unsigned*>(yt + (size_t)cc * ldt + r) = v;
        else yt[(size_t)cc * ldt + r] = (unsigned short)(v & 0xffffu);
      }
    }
  }
}
}  // namespace ts

/* ts_w2v_cast_bf16_t that also ADDS the column sums of x to colsum (f32 [c], may be NULL); see include/thunder_speech_amd.h */
extern "C" int ts_w2v_cast_bf16_t_colsum(const float* x, int64_t ldx, int64_t rows, int32_t c, void* y, int64_t ldy, void* yt, int64_t ldt, int64_t rows_pad,
                                         float* colsum, void* stream_) {
  if (!x || (!y && !yt) || rows <= 0 || c <= 0 || ldx < c || rows >= (1ll << 31)) return TS_EINVAL;
  if (y && ldy < c) return TS_EINVAL;
  if (yt && (rows_pad < rows || ldt < rows_pad || rows_pad >= (1ll << 31))) return TS_EINVAL;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const long long rp = yt ? rows_pad : rows;
  (void)hipGetLastError();
  const bool vec = c % 4 == 0 && ldx % 4 == 0 && al16(x) && (!y || (ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0)) &&
                   (!yt || (ldt % 2 == 0 && (reinterpret_cast<uintptr_t>(yt) & 3) == 0));
  if (vec)
    hipLaunchKernelGGL(ts::cast_bf16_t4_kernel, dim3((unsigned)((c + 63) / 64), (unsigned)((rp + 63) / 64)), dim3(256), 0, stream, x, (long long)ldx, (int)rows, (int)c,
                       static_cast<unsigned short*>(y), (long long)ldy, static_cast<unsigned short*>(yt), (long long)ldt, (int)rows_pad, colsum);
  else
  hipLaunchKernelGGL(ts::cast_bf16_t_kernel, dim3((unsigned)((c + 63) / 64), (unsigned)((rp + 63) / 64)), dim3(256), 0, stream, x, (long long)ldx, (int)rows, (int)c,
                     static_cast<unsigned short*>(y), (long long)ldy, static_cast<unsigned short*>(yt), (long long)ldt, (int)rows_pad, colsum);
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_w2v_cast_bf16_t(const float* x, int64_t ldx, int64_t rows, int32_t c, void* y, int64_t ldy, void* yt, int64_t ldt, int64_t rows_pad,
                                  void* stream_) {
  return ts_w2v_cast_bf16_t_colsum(x, ldx, rows, c, y, ldy, yt, ldt, rows_pad, nullptr, stream_);
}

extern "C" int64_t ts_w2v_layernorm_bwd_workspace(int64_t rows, int32_t c) {
  if (rows <= 0 || c <= 0) return TS_EINVAL;
  const long long waves = (rows < 4096 ? (rows + 3) / 4 * 4 : 4096);
  return (int64_t)waves * 2 * c * sizeof(float);
}

/* dx [rows][c]; dgamma, dbeta [c] are ADDED to (zero them, or pass the parameter's gradient); workspace: ts_w2v_layernorm_bwd_workspace bytes */
extern "C" int ts_w2v_layernorm_bwd(const float* x, const float* res, const float* gamma, const float* dy, float eps, int64_t rows, int32_t c, float* dx,
                                    float* dgamma, float* dbeta, void* workspace, void* stream_) {
  if (!x || !gamma || !dy || !dx || !dgamma || !dbeta || !workspace || rows <= 0 || c <= 0) return TS_EINVAL;
  if (c % 4 || c > 4096 || !al16(x) || !al16(dy) || !al16(dx) || !al16(gamma) || (res && !al16(res)) || !al16(workspace)) return TS_EUNSUPPORTED;
  TS_STREAM;
  const long long waves = (rows < 4096 ? (rows + 3) / 4 * 4 : 4096);
  float* part = static_cast<float*>(workspace);
  const dim3 grid((unsigned)(waves / 4));
#define TS_LNB(NV_) hipLaunchKernelGGL(ln_bwd_kernel<NV_>, grid, dim3(256), 0, stream, x, res, gamma, dy, dx, part, (long long)rows, c, eps)
  if (c <= 512) TS_LNB(2); else if (c <= 1024) TS_LNB(4); else if (c <= 2048) TS_LNB(8); else TS_LNB(16);
#undef TS_LNB
  // partials [waves][2][c]: even rows are d gamma, odd rows d beta
  const dim3 cg((c + 63) / 64, 16);
  hipLaunchKernelGGL(colsum_kernel, cg, dim3(256), 0, stream, part, dgamma, waves, c, (long long)2 * c);
  hipLaunchKernelGGL(colsum_kernel, cg, dim3(256), 0, stream, part + c, dbeta, waves, c, (long long)2 * c);
  return hip_status(hipGetLastError());
}

namespace ts {
// dgamma[j] / dbeta[j] += sums over the partial rows of ts_w2v_layernorm_bwd's workspace [waves][2][c] (blockIdx.z: 0 = dgamma, 1 = dbeta; blockIdx.y: a slice of
// the rows): ONE launch for both instead of two ts_w2v_colsum launches; the destinations were zeroed by ln_bwd_kernel (no fill launches).  (A single block per
// column group with plain stores was built first: 1 024 dependent-latency loads per thread, 65 us per launch instead of 12.)
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta, long long waves, int c) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const float* const p = part + (size_t)blockIdx.z * c;
  float s = 0.f;
  if (col < c)
    for (long long r = (long long)blockIdx.y * 4 + rg; r < waves; r += (long long)gridDim.y * 4) s += p[r * 2 * c + col];
  __shared__ float sm[4][64];
  sm[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && col < c) atomicAdd((blockIdx.z ? dbeta : dgamma) + col, (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]));
}
}  // namespace ts

/* ts_w2v_layernorm_bwd with dgamma / dbeta WRITTEN (not added to): no zero fill by the caller, one reducing launch for both; see include/thunder_speech_amd.h */
extern "C" int ts_w2v_layernorm_bwd_set(const float* x, const float* res, const float* gamma, const float* dy, float eps, int64_t rows, int32_t c, float* dx,
                                        float* dgamma, float* dbeta, void* workspace, void* stream_) {
  if (!x || !gamma || !dy || !dx || !dgamma || !dbeta || !workspace || rows <= 0 || c <= 0) return TS_EINVAL;
  if (c % 4 || c > 4096 || !al16(x) || !al16(dy) || !al16(dx) || !al16(gamma) || (res && !al16(res)) || !al16(workspace)) return TS_EUNSUPPORTED;
  TS_STREAM;
  const long long waves = (rows < 4096 ? (rows + 3) / 4 * 4 : 4096);
  float* part = static_cast<float*>(workspace);
  const dim3 grid((unsigned)(waves / 4));
#define TS_LNB(NV_) hipLaunchKernelGGL(ln_bwd_kernel<NV_>, grid, dim3(256), 0, stream, x, res, gamma, dy, dx, part, (long long)rows, c, eps, dgamma, dbeta)
  if (c <= 512) TS_LNB(2); else if (c <= 1024) TS_LNB(4); else if (c <= 2048) TS_LNB(8); else TS_LNB(16);
#undef TS_LNB
  hipLaunchKernelGGL(ts::ln_bwd_reduce_kernel, dim3((c + 63) / 64, 16, 2), dim3(256), 0, stream, part, dgamma, dbeta, waves, c);
  return hip_status(hipGetLastError());
}

/* out[j] += sum_r x[r * ld + j], j < c */
extern "C" int ts_w2v_colsum(const float* x, int64_t rows, int32_t c, int64_t ld, float* out, void* stream_) {
  if (!x || !out || rows <= 0 || c <= 0 || ld < c) return TS_EINVAL;
  TS_STREAM;
  const unsigned gy = (unsigned)(rows >= 4096 ? 256 : (rows + 15) / 16);
  hipLaunchKernelGGL(colsum_kernel, dim3((c + 63) / 64, gy ? gy : 1), dim3(256), 0, stream, x, out, (long long)rows, c, (long long)ld);
  return hip_status(hipGetLastError());
}

/* y = gelu(z + bias[col]) over [rows][c] (n = rows * c elements); bias may be NULL */
extern "C" int ts_w2v_gelu_fwd(const float* z, const float* bias, int32_t c, float* y, int64_t n, void* stream_) {
  if (!z || !y || n <= 0 || (bias && (c <= 0 || n % c))) return TS_EINVAL;
  if (!al16(z) || !al16(y) || (bias && (c % 4 || !al16(bias)))) return TS_EUNSUPPORTED;
  TS_STREAM;
  hipLaunchKernelGGL(gelu_kernel, dim3(blk4(n)), dim3(256), 0, stream, z, bias, c, (const float*)nullptr, y, (long long)n);
  return hip_status(hipGetLastError());
}

extern "C" int ts_w2v_gelu_bwd(const float* z, const float* bias, int32_t c, const float* dy, float* dz, int64_t n, void* stream_) {
  if (!z || !dy || !dz || n <= 0 || (bias && (c <= 0 || n % c))) return TS_EINVAL;
  if (!al16(z) || !al16(dy) || !al16(dz) || (bias && (c % 4 || !al16(bias)))) return TS_EUNSUPPORTED;
  TS_STREAM;
  hipLaunchKernelGGL(gelu_kernel, dim3(blk4(n)), dim3(256), 0, stream, z, bias, c, dy, dz, (long long)n);
  return hip_status(hipGetLastError());
}

/* s f32 [batch][heads][t][pitch] scores -> probabilities in place; key_len int32 [batch] or NULL */
extern "C" int ts_w2v_softmax_fwd(float* s, const int32_t* key_len, int32_t batch, int32_t heads, int32_t t, int32_t pitch, float scale, void* stream_) {
  if (!s || batch <= 0 || heads <= 0 || t <= 0 || pitch < t) return TS_EINVAL;
  TS_STREAM;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)(((long long)heads * t + 3) / 4), batch), dim3(256), 0, stream, s, key_len, heads, t, pitch, scale);
  return hip_status(hipGetLastError());
}

/* dp [rows][pitch] -> d scores in place: scale p (dp - <dp, p>) */
extern "C" int ts_w2v_softmax_bwd(const float* p, float* dp, int64_t rows, int32_t t, int32_t pitch, float scale, void* stream_) {
  if (!p || !dp || rows <= 0 || t <= 0 || pitch < t) return TS_EINVAL;
  TS_STREAM;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, p, dp, (long long)rows, t, pitch, scale);
  return hip_status(hipGetLastError());
}

/* extract = 0: dst [batch][t_dst][c] = src [batch][t][c] moved down by `left` rows, zeros around; 1: dst [batch][t][c] = rows left .. left + t of src */
extern "C" int ts_w2v_pad_rows(const float* src, float* dst, int32_t batch, int32_t t, int32_t t_dst, int32_t left, int32_t c, int32_t extract,
                               void* stream_) {
  if (!src || !dst || batch <= 0 || t <= 0 || t_dst < t + left || left < 0 || c <= 0) return TS_EINVAL;
  if (c % 4 || !al16(src) || !al16(dst)) return TS_EUNSUPPORTED;
  TS_STREAM;
  const long long n = (long long)(extract ? t : t_dst) * c;
  hipLaunchKernelGGL(pad_rows_kernel, dim3(blk4(n), batch), dim3(256), 0, stream, src, dst, t, t_dst, left, c, extract);
  return hip_status(hipGetLastError());
}

/* forward (dembed NULL): x[r][:] = embed for every row with mask[r] != 0.  backward (embed NULL): x is dy: dembed += its masked rows, which become 0 */
extern "C" int ts_w2v_mask_embed(float* x, const uint8_t* mask, const float* embed, float* dembed, int64_t rows, int32_t c, void* stream_) {
  if (!x || !mask || (!embed == !dembed) || rows <= 0 || c <= 0) return TS_EINVAL;
  if (c % 4 || !al16(x) || (embed && !al16(embed))) return TS_EUNSUPPORTED;
  TS_STREAM;
  hipLaunchKernelGGL(mask_embed_kernel, dim3(blk4((long long)rows * c)), dim3(256), 0, stream, x, mask, embed, dembed, (long long)rows, c);
  return hip_status(hipGetLastError());
}

extern "C" int ts_w2v_add(const float* a, const float* b, float* y, int64_t n, void* stream_) {
  if (!a || !b || !y || n <= 0) return TS_EINVAL;
  if (!al16(a) || !al16(b) || !al16(y)) return TS_EUNSUPPORTED;
  TS_STREAM;
  hipLaunchKernelGGL(add_kernel, dim3(blk4(n)), dim3(256), 0, stream, a, b, y, (long long)n);
  return hip_status(hipGetLastError());
}
