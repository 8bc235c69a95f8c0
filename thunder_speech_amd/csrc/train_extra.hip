// Training-mode pieces of the encoder blocks beyond the QuartzNet defaults (same conventions as csrc/train_enc.hip: pitched
// activation rows, `act` = 0 f32 / 1 bf16 storage with f32 arithmetic):
//   strided 1x1 MaskedConv1d (residual branch of a strided block, quartznet/blocks.py:301-311, citrinet/blocks.py:156-165):
//     mask + subsample in one pass; the 1x1 conv itself is the pointwise GEMM that follows
//   SqueezeExcite (citrinet/blocks.py:70-83) forward and backward: the two passes over the activation each way; the
//     [B, C]-sized bottleneck (two bias-free linears, ReLU, sigmoid) and its backward are one / two small launches (se_gate_*_kernel)
//   nn.Dropout (quartznet/blocks.py:227-228, blocks.py:238): Philox mask re-drawn in the backward pass
#include "ts_common.hpp"
#include "ts_philox.hpp"

namespace ts {

typedef unsigned short bf16_t;
__device__ __forceinline__ float ldf(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ float ldf(const bf16_t* p, size_t i) { return bf16_to_f32(p[i]); }
__device__ __forceinline__ void stf(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void stf(bf16_t* p, size_t i, float v) { p[i] = (bf16_t)(pack_bf16(v, 0.f) & 0xffffu); }

__device__ __forceinline__ int clamp_len2(const int* len, int b, int t) {
  if (!len) return t;
  const int l = len[b];
  return l < 0 ? 0 : (l > t ? t : l);
}

// forward: y[row][j] = x[row][j*s] if j*s < len[b] else 0
template <class T>
__global__ __launch_bounds__(256) void subsample_fwd_kernel(const T* __restrict__ x, const int* __restrict__ len, T* __restrict__ y,
                                                             int ch, int t_in, int t_out, int stride, int pitch_in, int pitch_out) {
  const int row = blockIdx.x, b = row / ch;
  const int l = clamp_len2(len, b, t_in);
  for (int j = blockIdx.y * 1024 + threadIdx.x; j < t_out && j < (int)(blockIdx.y + 1) * 1024; j += 256) {
    const int ti = j * stride;
    stf(y, (size_t)row * pitch_out + j, ti < l ? ldf(x, (size_t)row * pitch_in + ti) : 0.f);
  }
}
// backward: dx[row][t] = dy[row][t/s] if t % s == 0 and t < len[b] and t/s < t_out else 0
template <class T>
__global__ __launch_bounds__(256) void subsample_bwd_kernel(const T* __restrict__ dy, const int* __restrict__ len, T* __restrict__ dx,
                                                             int ch, int t_in, int t_out, int stride, int pitch_in, int pitch_out) {
  const int row = blockIdx.x, b = row / ch;
  const int l = clamp_len2(len, b, t_in);
  for (int t = blockIdx.y * 1024 + threadIdx.x; t < t_in && t < (int)(blockIdx.y + 1) * 1024; t += 256) {
    const int j = t / stride;
    stf(dx, (size_t)row * pitch_in + t, (t % stride == 0 && t < l && j < t_out) ? ldf(dy, (size_t)row * pitch_out + j) : 0.f);
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one wave per (clip, channel) row.  MODE 0: out[row] = mean_t a[row][t];  MODE 1: out[row] = sum_t a[row][t] * b[row][t]
template <int MODE, class T>
__global__ __launch_bounds__(256) void se_row_reduce_kernel(const T* __restrict__ a, const T* __restrict__ b2, float* __restrict__ out,
                                                             long long rows, int t, int pitch) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int i = lane; i < t; i += 64) s += MODE ? ldf(a, row * pitch + i) * ldf(b2, row * pitch + i) : ldf(a, row * pitch + i);
  s = wave_sum(s);
  if (lane == 0) out[row] = MODE ? s : s / (float)t;
}

// y[row][t] = x[row][t] * g[row] + (add ? add[row] * inv_t : 0)
template <class T>
__global__ __launch_bounds__(256) void se_scale_kernel(const T* __restrict__ x, const float* __restrict__ g, const float* __restrict__ add,
                                                        float inv_t, T* __restrict__ y, int t, int pitch) {
  const long long row = blockIdx.x;
  const float gr = g[row], ar = add ? add[row] * inv_t : 0.f;
  for (int i = blockIdx.y * 1024 + threadIdx.x; i < t && i < (int)(blockIdx.y + 1) * 1024; i += 256)
    stf(y, row * pitch + i, fmaf(ldf(x, row * pitch + i), gr, ar));
}

// ---- the [B, C] bottleneck of SqueezeExcite (citrinet/blocks.py:72-83: Linear(C, C/r, bias=False) -> ReLU -> Linear(C/r, C, bias=False) -> sigmoid) ----
// One workgroup of 16 waves per clip.  hid[j] = relu(sum_i w1[j][i] mean[i]): a wave per output, lanes over the contraction (coalesced rows);
// gate[c] = sigmoid(sum_j w2[c][j] hid[j]): 16 lanes per output (the contraction is only C/r long), 4 outputs per wave at a time.
__global__ __launch_bounds__(1024) void se_gate_kernel(const float* __restrict__ mean, const float* __restrict__ w1, const float* __restrict__ w2,
                                                        float* __restrict__ hid_out, float* __restrict__ gate, int channels, int hidden) {
  extern __shared__ float se_sm[];
  float* xm = se_sm;
  float* h = se_sm + channels;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < channels; i += 1024) xm[i] = mean[(size_t)b * channels + i];
  __syncthreads();
  // 32 workgroups on 256 CUs: the kernel is latency-bound, so every wave keeps EIGHT outputs' loads in flight at a time (one output after
  // the other cost ~70 us per launch at 1024 / 128 channels, measured in the Citrinet step)
  for (int j0 = wave * 8; j0 < hidden; j0 += 128) {
    float s[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) s[o] = 0.f;
    for (int i = lane; i < channels; i += 64) {
      const float x = xm[i];
#pragma unroll
      for (int o = 0; o < 8; ++o) s[o] = fmaf(j0 + o < hidden ? w1[(size_t)(j0 + o) * channels + i] : 0.f, x, s[o]);
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const float v = fmaxf(wave_sum(s[o]), 0.f);
      if (lane == 0 && j0 + o < hidden) {
        h[j0 + o] = v;
        if (hid_out) hid_out[(size_t)b * hidden + j0 + o] = v;
      }
    }
  }
  __syncthreads();
  const int sub = lane & 15, q = lane >> 4;
  for (int c0 = wave * 16; c0 < channels; c0 += 256) {          // a 16-lane group: outputs c0 + q, + 4, + 8, + 12 at once
    float s[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) s[o] = 0.f;
    for (int j = sub; j < hidden; j += 16) {
      const float hv = h[j];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int c = c0 + q + 4 * o;
        s[o] = fmaf(c < channels ? w2[(size_t)c * hidden + j] : 0.f, hv, s[o]);
      }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float v = s[o];
#pragma unroll
      for (int m = 8; m > 0; m >>= 1) v += __shfl_xor(v, m);
      const int c = c0 + q + 4 * o;
      if (sub == 0 && c < channels) gate[(size_t)b * channels + c] = 1.f / (1.f + __expf(-v));
    }
  }
}

// backward through the bottleneck, per clip: dz = dgate * g (1 - g); dhid[j] = (hid[j] > 0) sum_c dz[c] w2[c][j]; dmean[i] = sum_j dhid[j] w1[j][i].
// dz and dhid are kept ([B][C], [B][hidden]) for the weight gradients, which sum over the clips (se_gate_wgrad_kernel).
__global__ __launch_bounds__(1024) void se_gate_bwd_kernel(const float* __restrict__ dgate, const float* __restrict__ gate, const float* __restrict__ hid,
                                                            const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ dz_out,
                                                            float* __restrict__ dhid_out, float* __restrict__ dmean, int channels, int hidden) {
  extern __shared__ float se_sm[];
  float* dz = se_sm;                       // [channels]
  float* dh = se_sm + channels;            // [hidden]
  float* part = dh + hidden;               // [1024]
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < channels; c += 1024) {
    const float g = gate[(size_t)b * channels + c];
    const float v = dgate[(size_t)b * channels + c] * g * (1.f - g);
    dz[c] = v;
    dz_out[(size_t)b * channels + c] = v;
  }
  __syncthreads();
  int nj = 64;
  while (nj < hidden && nj < 1024) nj <<= 1;             // threads = slices x nj: coalesced rows of w2, slices over the channels
  const int slices = 1024 / nj, jl = tid % nj, cs = tid / nj;
  for (int j0 = 0; j0 < hidden; j0 += nj) {
    const int j = j0 + jl;
    float s = 0.f;
    if (j < hidden)
      for (int c = cs; c < channels; c += slices) s = fmaf(dz[c], w2[(size_t)c * hidden + j], s);
    part[tid] = s;
    __syncthreads();
    if (cs == 0 && j < hidden) {
      for (int k = 1; k < slices; ++k) s += part[k * nj + jl];
      s = hid[(size_t)b * hidden + j] > 0.f ? s : 0.f;
      dh[j] = s;
      dhid_out[(size_t)b * hidden + j] = s;
    }
    __syncthreads();
  }
  for (int i = tid; i < channels; i += 1024) {
    float s = 0.f;
#pragma unroll 4
    for (int j = 0; j < hidden; ++j) s = fmaf(dh[j], w1[(size_t)j * channels + i], s);
    dmean[(size_t)b * channels + i] = s;
  }
}

// dw2[c][j] = sum_b dz[b][c] hid[b][j]  (first channels * hidden threads);  dw1[j][i] = sum_b dhid[b][j] mean[b][i]  (the rest)
__global__ __launch_bounds__(256) void se_gate_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ hid, const float* __restrict__ dhid,
                                                             const float* __restrict__ mean, float* __restrict__ dw1, float* __restrict__ dw2, int batch,
                                                             int channels, int hidden) {
  const long long n = (long long)channels * hidden;
  long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx < n) {
    const int c = (int)(idx / hidden), j = (int)(idx % hidden);
    float s = 0.f;
    for (int b = 0; b < batch; ++b) s = fmaf(dz[(size_t)b * channels + c], hid[(size_t)b * hidden + j], s);
    dw2[idx] = s;
  } else if (idx < 2 * n) {
    idx -= n;
    const int j = (int)(idx / channels), i = (int)(idx % channels);
    float s = 0.f;
    for (int b = 0; b < batch; ++b) s = fmaf(dhid[(size_t)b * hidden + j], mean[(size_t)b * channels + i], s);
    dw1[idx] = s;
  }
}

// element (row, i) draws word (e & 3) of Philox counter e >> 2, e = row * t + i: the mask depends on the LOGICAL index only.  A thread owns one
// Philox block = four consecutive logical elements (which may straddle a row end when t % 4 != 0: the address is formed per element).
template <class T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long long n_el, int t, int pitch, float p, float scale,
                                                       unsigned long long seed, const unsigned long long* __restrict__ nonce) {
  if (nonce) seed += *nonce;                             // replay counter of a captured training step (hipGraph): a new mask per replay
  const unsigned long long e0 = 4ull * ((unsigned long long)blockIdx.x * 256 + threadIdx.x);
  if (e0 >= (unsigned long long)n_el) return;
  const Philox4 r = philox(seed, PHILOX_DROPOUT, e0 >> 2);
  long long row = (long long)(e0 / (unsigned)t);
  int i = (int)(e0 - (unsigned long long)row * t);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (e0 + q < (unsigned long long)n_el) {
      const size_t at = (size_t)row * pitch + i;
      stf(y, at, u01(r.v[q]) >= p ? ldf(x, at) * scale : 0.f);
      if (++i == t) { i = 0; ++row; }
    }
  }
}

// the same masks four elements at a time (f32 rows, t and pitch multiples of 4, 16-byte aligned rows): element e = row * t + i with i % 4 == 0 starts
// Philox block e >> 2, whose four words are the draws of elements i .. i + 3 -- ONE block and one 16-byte load / store per thread instead of four
// blocks and four scalar accesses (wav2vec2 fine-tuning: 196 launches x 57 us -> see profiles/round6_c5_finetune.md)
__global__ __launch_bounds__(256) void dropout4_kernel(const float* __restrict__ x, float* __restrict__ y, int t, int pitch, float p, float scale,
                                                       unsigned long long seed, const unsigned long long* __restrict__ nonce) {
  const long long row = blockIdx.x;
  if (nonce) seed += *nonce;
  const int i = blockIdx.y * 1024 + threadIdx.x * 4;
  if (i >= t) return;
  const unsigned long long e = (unsigned long long)row * t + i;
  const Philox4 r = philox(seed, PHILOX_DROPOUT, e >> 2);
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + row * pitch + i);
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = u01(r.v[j]) >= p ? v[j] * scale : 0.f;
  *reinterpret_cast<f32x4*>(y + row * pitch + i) = o;
}

static inline dim3 rgrid(long long rows, int t) { return dim3((unsigned)rows, (unsigned)((t + 1023) / 1024)); }

}  // namespace ts

using namespace ts;
#define TS_ACT(act, expr_f32, expr_bf16) do { if (act) { expr_bf16; } else { expr_f32; } } while (0)

extern "C" int ts_train_subsample_mask(const void* x, const int32_t* len, void* y, int32_t batch, int32_t ch, int32_t t_in,
                                       int32_t t_out, int32_t stride, int32_t backward, int32_t pitch_in, int32_t pitch_out, int32_t act,
                                       void* stream_) {
  if (!x || !y || batch <= 0 || ch <= 0 || t_in <= 0 || t_out <= 0 || stride < 1 || (t_out - 1) * stride >= t_in) return TS_EINVAL;
  if (pitch_in < t_in || pitch_out < t_out || act < 0 || act > 1) return TS_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();
  const long long rows = (long long)batch * ch;
  if (!backward)
    TS_ACT(act,
           hipLaunchKernelGGL(subsample_fwd_kernel<float>, rgrid(rows, t_out), dim3(256), 0, stream, (const float*)x, len, (float*)y, ch, t_in, t_out, stride, pitch_in, pitch_out),
           hipLaunchKernelGGL(subsample_fwd_kernel<bf16_t>, rgrid(rows, t_out), dim3(256), 0, stream, (const bf16_t*)x, len, (bf16_t*)y, ch, t_in, t_out, stride, pitch_in, pitch_out));
  else
    TS_ACT(act,
           hipLaunchKernelGGL(subsample_bwd_kernel<float>, rgrid(rows, t_in), dim3(256), 0, stream, (const float*)x, len, (float*)y, ch, t_in, t_out, stride, pitch_in, pitch_out),
           hipLaunchKernelGGL(subsample_bwd_kernel<bf16_t>, rgrid(rows, t_in), dim3(256), 0, stream, (const bf16_t*)x, len, (bf16_t*)y, ch, t_in, t_out, stride, pitch_in, pitch_out));
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_se_pool(const void* x, float* mean, int64_t rows, int32_t t, int32_t pitch, int32_t act, void* stream) {
  if (!x || !mean || rows <= 0 || t <= 0 || pitch < t || act < 0 || act > 1) return TS_EINVAL;
  (void)hipGetLastError();
  const dim3 grid((unsigned)((rows + 3) / 4));
  TS_ACT(act,
         hipLaunchKernelGGL((se_row_reduce_kernel<0, float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)nullptr, mean, (long long)rows, t, pitch),
         hipLaunchKernelGGL((se_row_reduce_kernel<0, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)nullptr, mean, (long long)rows, t, pitch));
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_se_rowdot(const void* a, const void* b, float* out, int64_t rows, int32_t t, int32_t pitch, int32_t act, void* stream) {
  if (!a || !b || !out || rows <= 0 || t <= 0 || pitch < t || act < 0 || act > 1) return TS_EINVAL;
  (void)hipGetLastError();
  const dim3 grid((unsigned)((rows + 3) / 4));
  TS_ACT(act,
         hipLaunchKernelGGL((se_row_reduce_kernel<1, float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, out, (long long)rows, t, pitch),
         hipLaunchKernelGGL((se_row_reduce_kernel<1, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, out, (long long)rows, t, pitch));
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_se_scale(const void* x, const float* gate, const float* add_mean, void* y, int64_t rows, int32_t t, int32_t pitch,
                                 int32_t act, void* stream) {
  if (!x || !gate || !y || rows <= 0 || t <= 0 || pitch < t || act < 0 || act > 1) return TS_EINVAL;
  (void)hipGetLastError();
  TS_ACT(act,
         hipLaunchKernelGGL(se_scale_kernel<float>, rgrid(rows, t), dim3(256), 0, (hipStream_t)stream, (const float*)x, gate, add_mean, 1.0f / (float)t, (float*)y, t, pitch),
         hipLaunchKernelGGL(se_scale_kernel<bf16_t>, rgrid(rows, t), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, gate, add_mean, 1.0f / (float)t, (bf16_t*)y, t, pitch));
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_se_gate_fwd(const float* mean, const float* w1, const float* w2, float* hid, float* gate, int32_t batch, int32_t channels,
                                    int32_t hidden, void* stream) {
  if (!mean || !w1 || !w2 || !gate || batch <= 0 || channels <= 0 || hidden <= 0) return TS_EINVAL;
  const size_t lds = (size_t)(channels + hidden) * sizeof(float);
  if (lds > 64 * 1024) return TS_EUNSUPPORTED;
  (void)hipGetLastError();
  hipLaunchKernelGGL(se_gate_kernel, dim3(batch), dim3(1024), lds, (hipStream_t)stream, mean, w1, w2, hid, gate, channels, hidden);
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_se_gate_bwd(const float* dgate, const float* gate, const float* hid, const float* mean, const float* w1, const float* w2,
                                    float* dz_ws, float* dhid_ws, float* dmean, float* dw1, float* dw2, int32_t batch, int32_t channels, int32_t hidden,
                                    void* stream) {
  if (!dgate || !gate || !hid || !mean || !w1 || !w2 || !dz_ws || !dhid_ws || !dmean || !dw1 || !dw2) return TS_EINVAL;
  if (batch <= 0 || channels <= 0 || hidden <= 0) return TS_EINVAL;
  const size_t lds = (size_t)(channels + hidden + 1024) * sizeof(float);
  if (lds > 64 * 1024) return TS_EUNSUPPORTED;
  (void)hipGetLastError();
  hipLaunchKernelGGL(se_gate_bwd_kernel, dim3(batch), dim3(1024), lds, (hipStream_t)stream, dgate, gate, hid, w1, w2, dz_ws, dhid_ws, dmean, channels, hidden);
  const long long n = 2ll * channels * hidden;
  hipLaunchKernelGGL(se_gate_wgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dz_ws, hid, dhid_ws, mean, dw1, dw2, batch,
                     channels, hidden);
  return hip_status(hipGetLastError());
}

__global__ void counter_add_kernel(unsigned long long* c, unsigned long long inc) { *c += inc; }

extern "C" int ts_counter_add(uint64_t* counter, uint64_t inc, void* stream) {
  if (!counter) return TS_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)counter, (unsigned long long)inc);
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_dropout(const void* x, void* y, int64_t rows, int32_t t, int32_t pitch, float p, uint64_t seed, const uint64_t* nonce,
                                int32_t act, void* stream) {
  if (!x || !y || rows <= 0 || t <= 0 || pitch < t || !(p >= 0.f) || p > 1.f || act < 0 || act > 1) return TS_EINVAL;
  const float scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
  (void)hipGetLastError();
  if (act == 0 && t % 4 == 0 && pitch % 4 == 0 && !(reinterpret_cast<uintptr_t>(x) & 15) && !(reinterpret_cast<uintptr_t>(y) & 15)) {
    hipLaunchKernelGGL(dropout4_kernel, rgrid(rows, t), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, t, pitch, p, scale, (unsigned long long)seed,
                       (const unsigned long long*)nonce);
    return hip_status(hipGetLastError());
  }
  const long long n_el = (long long)rows * t;
  const dim3 fgrid((unsigned)((n_el + 1023) / 1024));
  TS_ACT(act,
         hipLaunchKernelGGL(dropout_kernel<float>, fgrid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, n_el, t, pitch, p, scale, (unsigned long long)seed,
                            (const unsigned long long*)nonce),
         hipLaunchKernelGGL(dropout_kernel<bf16_t>, fgrid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, n_el, t, pitch, p, scale, (unsigned long long)seed,
                            (const unsigned long long*)nonce));
  return hip_status(hipGetLastError());
}
