// Micro-benchmark: VALU issue rates on gfx950 for the candidate depthwise-FIR inner ops.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
// Each kernel runs ITERS x 16 independent ops per lane; reports G lane-ops/s over the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

constexpr int ITERS = 4096;

#define BODY16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

__global__ void k_fma(float* out, const float* in) {
  float a = in[threadIdx.x], b = in[threadIdx.x + 64];
  float acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = in[i];
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    BODY16(OP)
#undef OP
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fmac_sgpr(float* out, const float* in) {
  float a = in[threadIdx.x];
  float sb = in[blockIdx.x & 7];
  sb = __builtin_amdgcn_readfirstlane(sb);
  float acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = in[i];
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(acc[i]) : "v"(a), "s"(sb));
    BODY16(OP)
#undef OP
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k_pkfma32(float* out, const float* in) {
  f32x2 a = {in[threadIdx.x], in[threadIdx.x+1]}, b = {in[threadIdx.x + 64], in[threadIdx.x+65]};
  f32x2 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x2{in[i], in[i+1]};
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    BODY16(OP)
#undef OP
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_dot2c(float* out, const float* in) {
  unsigned a = __float_as_uint(in[threadIdx.x]), b = __float_as_uint(in[threadIdx.x + 64]);
  float acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = in[i];
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    BODY16(OP)
#undef OP
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_dot2(float* out, const float* in) {
  unsigned a = __float_as_uint(in[threadIdx.x]), b = __float_as_uint(in[threadIdx.x + 64]);
  float acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = in[i];
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    BODY16(OP)
#undef OP
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_dot2_f16(float* out, const float* in) {
  unsigned a = __float_as_uint(in[threadIdx.x]), b = __float_as_uint(in[threadIdx.x + 64]);
  float acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = in[i];
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    BODY16(OP)
#undef OP
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_pkfma16(float* out, const float* in) {
  unsigned a = __float_as_uint(in[threadIdx.x]), b = __float_as_uint(in[threadIdx.x + 64]);
  unsigned acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = __float_as_uint(in[i]);
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    BODY16(OP)
#undef OP
  }
  unsigned s = 0; for (int i = 0; i < 16; ++i) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(s);
}
__global__ void k_perm(float* out, const float* in) {
  unsigned a = __float_as_uint(in[threadIdx.x]), b = __float_as_uint(in[threadIdx.x + 64]);
  unsigned acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = __float_as_uint(in[i]);
  unsigned sel = 0x07060302u;
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(sel));
    BODY16(OP)
#undef OP
  }
  unsigned s = 0; for (int i = 0; i < 16; ++i) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(s);
}
__global__ void k_lshl(float* out, const float* in) {
  unsigned acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = __float_as_uint(in[i]);
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(acc[i]));
    BODY16(OP)
#undef OP
  }
  unsigned s = 0; for (int i = 0; i < 16; ++i) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(s);
}
template <typename K>
int run(const char* name, K kern, int wavesPerSimd, double opsPerLaneIter, float* out, float* in) {
  int block = 256;                       // 4 waves = 1 per SIMD
  int grid = 256 * wavesPerSimd;         // per CU: wavesPerSimd blocks
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, out, in);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, out, in);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  double laneops = (double)grid * block * ITERS * 16 * opsPerLaneIter;
  // cycles per wave-instruction per SIMD at 2.4 GHz (nominal)
  double instr_per_simd = (double)wavesPerSimd * ITERS * 16;
  double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
  printf("%-14s waves/SIMD=%d  %8.3f ms  %9.1f G lane-ops/s  ~%.2f cyc/wave-instr/SIMD @2.4GHz\n", name, wavesPerSimd, ms, laneops / ms * 1e-6, cyc);
  return 0;
}

int main() {
  float *in, *out;
  CK(hipMalloc(&in, 1 << 20)); CK(hipMalloc(&out, 256 * 8 * 256 * 4 * 4));
  std::vector<float> h(1 << 18);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0.5f + 1e-3f * (i % 97);
  CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (int w : {1, 2, 4}) {
    run("v_fma_f32", k_fma, w, 1, out, in);
    run("v_fmac_sgpr", k_fmac_sgpr, w, 1, out, in);
    run("v_pk_fma_f32", k_pkfma32, w, 2, out, in);
    run("v_dot2c_bf16", k_dot2c, w, 2, out, in);
    run("v_dot2_bf16", k_dot2, w, 2, out, in);
    run("v_dot2_f16", k_dot2_f16, w, 2, out, in);
    run("v_pk_fma_f16", k_pkfma16, w, 2, out, in);
    run("v_perm_b32", k_perm, w, 1, out, in);
    run("v_lshlrev", k_lshl, w, 1, out, in);
  }
  return 0;
}
