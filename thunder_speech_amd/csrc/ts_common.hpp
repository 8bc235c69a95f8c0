// Shared device helpers for the thunder_speech_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "thunder_speech_amd.h"

namespace ts {

typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

#define TS_LDS __attribute__((address_space(3)))

// float pair -> packed bf16 pair (round-to-nearest-even).  Compiler-scheduled form: safe right after an MFMA (hipcc pads
// the MFMA -> VALU hazard), but it lowers to two single conversions plus a v_perm_b32.
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
// ONE v_cvt_pk_bf16_f32.  hipcc inserts no wait states for inline asm, so the operands must NOT be the result of an
// MFMA issued within the last ~20 cycles (cdna_hip_programming.md 5.7 item 2); used by the epilogues only.
__device__ __forceinline__ unsigned pack_bf16_settled(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// keep the first `nv` bf16 elements of an 8-element group, zero the rest (nv may be <=0 or >=8)
__device__ __forceinline__ u32x4 keep_first(u32x4 v, int nv) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = nv - 2 * j;
    const unsigned m = r >= 2 ? 0xFFFFFFFFu : (r == 1 ? 0x0000FFFFu : 0u);
    v[j] &= m;
  }
  return v;
}

__host__ __device__ constexpr int round_up(int x, int m) { return (x + m - 1) / m * m; }

inline int hip_status(hipError_t e) { return e == hipSuccess ? TS_OK : (int)e; }

}  // namespace ts
