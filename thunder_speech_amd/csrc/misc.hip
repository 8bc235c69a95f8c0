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

}  // namespace ts

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
