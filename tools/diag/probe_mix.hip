// Probe: do v_mfma_f32_32x32x16_bf16 (consumers) and v_mfma_f32_4x4x4_16b_bf16 (producers) share a SIMD's matrix core without
// loss?  One workgroup of 768 threads per CU = 12 waves = 3 per SIMD, exactly the split kernel's shape: waves 0-7 issue the big
// MFMAs (24 per "stage"), waves 8-11 the small ones (108 per "stage"); no memory, no LDS, optional barrier per stage.
//   hipcc --offload-arch=gfx950 -O3 -o probe_mix probe_mix.hip && ./probe_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
constexpr int STAGES = 4096;

template <int MODE, bool BARRIER>      // MODE 1: consumers only, 2: producers only, 3: both
__global__ __launch_bounds__(768) void k_mix(float* o, const s16x8* a8, const s16x4* a4) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float r = 0.f;
  if (wave < 8) {
    s16x8 av = a8[lane], bv = a8[lane + 64];
    f32x16 c[6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
    for (int s = 0; s < STAGES; ++s) {
      if (MODE & 1) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int i = 0; i < 6; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c[i], 0, 0, 0);
      }
      if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    for (int i = 0; i < 6; ++i) r += c[i][0] + c[i][7];
  } else {
    s16x4 av = a4[lane], bv = a4[lane + 64];
    f32x4 d[6];
    for (int i = 0; i < 6; ++i) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < STAGES; ++s) {
      if (MODE & 2) {
#pragma unroll
        for (int kk = 0; kk < 18; ++kk)
#pragma unroll
          for (int i = 0; i < 6; ++i) d[i] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(av, bv, d[i], 0, 0, 0);
      }
      if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    for (int i = 0; i < 6; ++i) r += d[i][0] + d[i][3];
  }
  o[blockIdx.x * 768 + threadIdx.x] = r;
}

template <typename K> static int run(const char* name, K kern, float* o, s16x8* a8, s16x4* a4) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(256), dim3(768), 0, 0, o, a8, a4); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(256), dim3(768), 0, 0, o, a8, a4); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-42s %8.3f ms  -> %7.1f ns per stage\n", name, ms, ms * 1e6 / STAGES);
  return 0;
}

int main() {
  float* o; s16x8* a8; s16x4* a4;
  CK(hipMalloc(&o, 256 * 768 * 4)); CK(hipMalloc(&a8, 128 * 16)); CK(hipMalloc(&a4, 128 * 8));
  CK(hipMemset(a8, 0, 128 * 16)); CK(hipMemset(a4, 0, 128 * 8));
  run("32x32x16 only (2 waves/SIMD x 24)", k_mix<1, false>, o, a8, a4);
  run("4x4x4 only (1 wave/SIMD x 108)", k_mix<2, false>, o, a8, a4);
  run("both, no barrier", k_mix<3, false>, o, a8, a4);
  run("32x32x16 only + barrier per stage", k_mix<1, true>, o, a8, a4);
  run("4x4x4 only + barrier per stage", k_mix<2, true>, o, a8, a4);
  run("both + barrier per stage", k_mix<3, true>, o, a8, a4);
  return 0;
}
