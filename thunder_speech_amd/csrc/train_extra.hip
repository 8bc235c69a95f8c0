// Training-mode pieces of the encoder blocks that the first training path left out (fp32 [B][C][T], one entry point per op
// and direction like csrc/train_enc.hip):
//   strided 1x1 MaskedConv1d (residual branch of a strided block, quartznet/blocks.py:301-311, citrinet/blocks.py:156-165):
//     mask + subsample in one pass; the 1x1 conv itself is the pointwise GEMM that follows
//   SqueezeExcite (citrinet/blocks.py:70-83) forward and backward: the two passes over the activation each way; the
//     [B, C]-sized bottleneck (two bias-free linears, ReLU, sigmoid) is a handful of tiny GEMMs done by the caller
#include "ts_common.hpp"

namespace ts {

__device__ __forceinline__ int clamp_len2(const int* len, int b, int t) {
  if (!len) return t;
  const int l = len[b];
  return l < 0 ? 0 : (l > t ? t : l);
}

// forward: y[b,c,j] = x[b,c,j*s] if j*s < len[b] else 0        (x [.., t_in], y [.., t_out])
__global__ __launch_bounds__(256) void subsample_fwd_kernel(const float* __restrict__ x, const int* __restrict__ len, float* __restrict__ y,
                                                             int batch, int ch, int t_in, int t_out, int stride) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)batch * ch * t_out) return;
  const int j = (int)(idx % t_out);
  const long long row = idx / t_out;
  const int b = (int)(row / ch);
  const int ti = j * stride;
  y[idx] = (ti < clamp_len2(len, b, t_in)) ? x[row * t_in + ti] : 0.f;
}
// backward: dx[b,c,t] = dy[b,c,t/s] if t % s == 0 and t < len[b] and t/s < t_out else 0
__global__ __launch_bounds__(256) void subsample_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ len, float* __restrict__ dx,
                                                             int batch, int ch, int t_in, int t_out, int stride) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)batch * ch * t_in) return;
  const int t = (int)(idx % t_in);
  const long long row = idx / t_in;
  const int b = (int)(row / ch);
  const int j = t / stride;
  dx[idx] = (t % stride == 0 && t < clamp_len2(len, b, t_in) && j < t_out) ? dy[row * t_out + j] : 0.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one wave per (clip, channel) row.  MODE 0: out[row] = mean_t a[row][t];  MODE 1: out[row] = sum_t a[row][t] * b[row][t]
template <int MODE>
__global__ __launch_bounds__(256) void se_row_reduce_kernel(const float* __restrict__ a, const float* __restrict__ b2, float* __restrict__ out,
                                                             long long rows, int t) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* pa = a + row * t;
  const float* pb = MODE ? b2 + row * t : nullptr;
  float s = 0.f;
  for (int i = lane; i < t; i += 64) s += MODE ? pa[i] * pb[i] : pa[i];
  s = wave_sum(s);
  if (lane == 0) out[row] = MODE ? s : s / (float)t;
}

// y[row][t] = x[row][t] * g[row] + (add ? add[row] * inv_t : 0)
__global__ __launch_bounds__(256) void se_scale_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ add,
                                                        float inv_t, float* __restrict__ y, long long rows, int t) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * t) return;
  const long long row = idx / t;
  y[idx] = fmaf(x[idx], g[row], add ? add[row] * inv_t : 0.f);
}

static inline unsigned nblocks(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace ts

extern "C" int ts_train_subsample_mask(const float* x, const int32_t* len, float* y, int32_t batch, int32_t ch, int32_t t_in,
                                       int32_t t_out, int32_t stride, int32_t backward, void* stream) {
  using namespace ts;
  if (!x || !y || batch <= 0 || ch <= 0 || t_in <= 0 || t_out <= 0 || stride < 1 || (t_out - 1) * stride >= t_in) return TS_EINVAL;
  (void)hipGetLastError();
  if (!backward)
    hipLaunchKernelGGL(subsample_fwd_kernel, dim3(nblocks((long long)batch * ch * t_out)), dim3(256), 0, (hipStream_t)stream, x, len, y,
                       batch, ch, t_in, t_out, stride);
  else
    hipLaunchKernelGGL(subsample_bwd_kernel, dim3(nblocks((long long)batch * ch * t_in)), dim3(256), 0, (hipStream_t)stream, x, len, y,
                       batch, ch, t_in, t_out, stride);
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_se_pool(const float* x, float* mean, int64_t rows, int32_t t, void* stream) {
  using namespace ts;
  if (!x || !mean || rows <= 0 || t <= 0) return TS_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(se_row_reduce_kernel<0>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, nullptr, mean,
                     (long long)rows, t);
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_se_rowdot(const float* a, const float* b, float* out, int64_t rows, int32_t t, void* stream) {
  using namespace ts;
  if (!a || !b || !out || rows <= 0 || t <= 0) return TS_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(se_row_reduce_kernel<1>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, b, out,
                     (long long)rows, t);
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_se_scale(const float* x, const float* gate, const float* add_mean, float* y, int64_t rows, int32_t t,
                                 void* stream) {
  using namespace ts;
  if (!x || !gate || !y || rows <= 0 || t <= 0) return TS_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(se_scale_kernel, dim3(nblocks((long long)rows * t)), dim3(256), 0, (hipStream_t)stream, x, gate, add_mean,
                     1.0f / (float)t, y, (long long)rows, t);
  return hip_status(hipGetLastError());
}
