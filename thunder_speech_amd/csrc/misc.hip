// Library identification and boundary layout helpers (reference layout f32 [B][C][T] <-> NCT-p bf16).
#include "ts_common.hpp"

#ifndef TS_BUILD_TARGET
#define TS_BUILD_TARGET "gfx950"
#endif

namespace ts {

__global__ void pack_kernel(const float* __restrict__ src, const int* __restrict__ len, unsigned short* __restrict__ dst,
                            int rows, int channels, int t, int pitch) {
  // one thread per 8 output elements (16 B); columns >= min(t, len[b]) are written as zero (tail-zero invariant)
  const int groups = pitch >> 3;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)rows * groups) return;
  const int row = (int)(idx / groups), g = (int)(idx % groups);
  int lim = t;
  if (len) { const int l = len[row / channels]; lim = l < t ? (l < 0 ? 0 : l) : t; }
  const float* s = src + (size_t)row * t + g * 8;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (g * 8 + i < lim) ? s[i] : 0.f;
  u32x4 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
  *reinterpret_cast<u32x4*>(dst + (size_t)row * pitch + g * 8) = o;
}

__global__ void unpack_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, int rows, int t, int pitch) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)rows * t) return;
  const int row = (int)(idx / t), c = (int)(idx % t);
  dst[idx] = bf16_to_f32(src[(size_t)row * pitch + c]);
}

// out[i] = floor((in[i] + add) / div) + plus in the arithmetic of the input type (f32 lengths stay f32 operations, integer lengths
// use floor division), written in `out_kind` and, optionally, once more as int32 (what the kernels take).  kinds: 0 f32, 1 i64, 2 i32
__global__ void lengths_map_kernel(const void* __restrict__ in, int in_kind, void* __restrict__ out, int out_kind, int* __restrict__ out_i32,
                                   int n, long long add, long long div, long long plus) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r;
  if (in_kind == 0) {
    // f32 lengths stay f32 operations (torch.div(..., rounding_mode="floor") on a float tensor).  The library is built with
    // -ffast-math, which turns the division into a multiplication by the reciprocal: 156 / 3 then comes out just below 52 --
    // so the quotient is corrected against the exact products (lengths are small integers or halves: q * div is exact in f32)
    const float v = static_cast<const float*>(in)[i] + (float)add, dv = (float)div;
    float q = floorf(v / dv);
    if ((q + 1.f) * dv <= v) q += 1.f;
    if (q * dv > v) q -= 1.f;
    r = (double)(q + (float)plus);
  } else {
    const long long v = (in_kind == 1 ? static_cast<const long long*>(in)[i] : (long long)static_cast<const int*>(in)[i]) + add;
    long long q = v / div;
    if ((v % div != 0) && ((v < 0) != (div < 0))) --q;                      // floor, like torch.div(rounding_mode="floor")
    r = (double)(q + plus);
  }
  if (out) {
    if (out_kind == 0) static_cast<float*>(out)[i] = (float)r;
    else if (out_kind == 1) static_cast<long long*>(out)[i] = (long long)r;
    else static_cast<int*>(out)[i] = (int)r;
  }
  if (out_i32) out_i32[i] = (int)r;
}

// im2col along time for the convolutions that have no fused kernel of their own (dense K > 1, depthwise stride > 2):
// out[b][u * C + c][t] = (0 <= ti < len[b]) ? x[b][c][ti] : 0,  ti = t * stride + u * dil - pad, for t < t_out; 0 from t_out to the pitch.
// One thread per 8 output frames (16-byte stores); the reads are element-wise (stride / dilation / odd padding leave no alignment).
__global__ void im2col_time_kernel(const unsigned short* __restrict__ x, const int* __restrict__ len, unsigned short* __restrict__ out,
                                   int channels, int pitch_in, int t_in, int k, int stride, int dil, int pad, int t_out, int pitch_out,
                                   long long n_groups) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_groups) return;
  const int groups = pitch_out >> 3;
  const int g = (int)(idx % groups);
  const long long row = idx / groups;                       // (b * k + u) * channels + c
  const int c = (int)(row % channels);
  const int u = (int)((row / channels) % k);
  const int b = (int)(row / ((long long)channels * k));
  int lim = len ? len[b] : t_in;
  lim = lim < t_in ? (lim < 0 ? 0 : lim) : t_in;
  const unsigned short* src = x + ((size_t)b * channels + c) * pitch_in;
  unsigned short v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int t = g * 8 + i;
    const int ti = t * stride + u * dil - pad;
    v[i] = (t < t_out && ti >= 0 && ti < lim) ? src[ti] : (unsigned short)0;
  }
  u32x4 o = {(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16), (unsigned)v[4] | ((unsigned)v[5] << 16),
             (unsigned)v[6] | ((unsigned)v[7] << 16)};
  *reinterpret_cast<u32x4*>(out + (size_t)row * pitch_out + g * 8) = o;
}

}  // namespace ts

extern "C" int ts_im2col_time(const void* x, const int32_t* len, void* out, int32_t batch, int32_t channels, int32_t t_in, int32_t pitch_in,
                              int32_t kernel, int32_t stride, int32_t dilation, int32_t padding, int32_t t_out, int32_t pitch_out, void* stream) {
  if (!x || !out || batch <= 0 || channels <= 0 || t_in <= 0 || kernel <= 0 || stride <= 0 || dilation <= 0 || padding < 0 || t_out <= 0 ||
      pitch_out < t_out || pitch_out % 8 || pitch_in < t_in)
    return TS_EINVAL;
  const long long n = (long long)batch * kernel * channels * (pitch_out / 8);
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::im2col_time_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, len,
                     (unsigned short*)out, channels, pitch_in, t_in, kernel, stride, dilation, padding, t_out, pitch_out, n);
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_lengths_map(const void* in, int32_t in_kind, void* out, int32_t out_kind, int32_t* out_i32, int32_t n, int64_t add,
                              int64_t div, int64_t plus, void* stream) {
  if (!in || (!out && !out_i32) || n <= 0 || div == 0 || in_kind < 0 || in_kind > 2 || out_kind < 0 || out_kind > 2) return TS_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::lengths_map_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, in_kind, out, out_kind,
                     out_i32, n, (long long)add, (long long)div, (long long)plus);
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_abi_version(void) { return TS_ABI_VERSION; }
extern "C" const char* ts_build_target(void) { return TS_BUILD_TARGET; }

extern "C" int ts_pack_activation(const float* src, const int32_t* len, int32_t batch, int32_t channels, int32_t t, void* dst,
                                  int32_t pitch, void* stream) {
  if (!src || !dst || batch <= 0 || channels <= 0 || t <= 0 || pitch < t || pitch % 8) return TS_EINVAL;
  const long long n = (long long)batch * channels * (pitch / 8);
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, len,
                     (unsigned short*)dst, batch * channels, channels, t, pitch);
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_unpack_activation(const void* src, int32_t batch, int32_t channels, int32_t t, int32_t pitch, float* dst,
                                    void* stream) {
  if (!src || !dst || batch <= 0 || channels <= 0 || t <= 0 || pitch < t) return TS_EINVAL;
  const long long n = (long long)batch * channels * t;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)src, dst, batch * channels, t, pitch);
  return ts::hip_status(hipGetLastError());
}
