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
typedef __attribute__((ext_vector_type(4))) int i32x4;

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

// Raw buffer descriptor (gfx9 layout) over [p, p + bytes): base, stride 0, num_records, 32-bit data format.
__device__ __forceinline__ i32x4 raw_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  return i32x4{(int)(a & 0xffffffffu), (int)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
// Global -> LDS DMA of 16 bytes per lane (1 KiB per wave): lane L's 16 bytes at buffer offset voff + soff land at LDS byte
// address lds_dst + 16 L.  `after` is a register the DMA must be ordered behind: it is not read, it only pins the
// instruction below its producer.  hipcc orders the DMA against other MEMORY operations only, so without it the DMA
// that refills a tap slot floats above the MFMAs that consume the slot's old contents -- whose LDS reads may still be in
// flight -- and with an L2-hot source it lands first (measured: the taps of the last pass came from the next stage in
// 1-5 % of the waves).  Behind the last MFMA of the pass every read of the slot has returned.
// hipcc neither counts this operation in its vmcnt model nor orders LDS reads behind it: callers wait with vm_wait<>.
__device__ __forceinline__ void lds_dma16(i32x4 rsrc, const void* lds_dst, int voff, int soff, float after = 0.f) {
  const unsigned addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(TS_LDS const char*)lds_dst);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(addr), "v"(voff), "s"(rsrc), "s"(soff), "v"(after) : "memory");
}

__host__ __device__ constexpr int round_up(int x, int m) { return (x + m - 1) / m * m; }

inline int hip_status(hipError_t e) { return e == hipSuccess ? TS_OK : (int)e; }

// compute units of the current device (one process per GPU: looked up once)
inline int cu_count() {
  static int n_cu = 0;
  if (n_cu) return n_cu;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    return n_cu = prop.multiProcessorCount;
  return n_cu = 256;
}

}  // namespace ts
