// Training-mode pieces of the encoder blocks beyond the QuartzNet defaults (same conventions as csrc/train_enc.hip: pitched
// activation rows, `act` = 0 f32 / 1 bf16 storage with f32 arithmetic):
//   strided 1x1 MaskedConv1d (residual branch of a strided block, quartznet/blocks.py:301-311, citrinet/blocks.py:156-165):
//     mask + subsample in one pass; the 1x1 conv itself is the pointwise GEMM that follows
//   SqueezeExcite (citrinet/blocks.py:70-83) forward and backward: the two passes over the activation each way; the
//     [B, C]-sized bottleneck (two bias-free linears, ReLU, sigmoid) is a handful of tiny GEMMs done by the caller
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

// element (row, i) draws word (e & 3) of Philox counter e >> 2, e = row * t + i: the mask depends on the LOGICAL index only
template <class T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int t, int pitch, float p, float scale,
                                                       unsigned long long seed, const unsigned long long* __restrict__ nonce) {
  const long long row = blockIdx.x;
  if (nonce) seed += *nonce;                             // replay counter of a captured training step (hipGraph): a new mask per replay
  for (int i = blockIdx.y * 1024 + threadIdx.x; i < t && i < (int)(blockIdx.y + 1) * 1024; i += 256) {
    const unsigned long long e = (unsigned long long)row * t + i;
    const Philox4 r = philox(seed, PHILOX_DROPOUT, e >> 2);
    stf(y, row * pitch + i, u01(r.v[e & 3]) >= p ? ldf(x, row * pitch + i) * scale : 0.f);
  }
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
  TS_ACT(act,
         hipLaunchKernelGGL(dropout_kernel<float>, rgrid(rows, t), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, t, pitch, p, scale, (unsigned long long)seed,
                            (const unsigned long long*)nonce),
         hipLaunchKernelGGL(dropout_kernel<bf16_t>, rgrid(rows, t), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, t, pitch, p, scale, (unsigned long long)seed,
                            (const unsigned long long*)nonce));
  return hip_status(hipGetLastError());
}
