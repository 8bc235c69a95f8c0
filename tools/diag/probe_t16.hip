// Probe for the Toeplitz depthwise on v_mfma_f32_16x16x32_bf16 (round 3): one wave = 16 channels of a stage, a 192-frame
// tile = 12 segments of 16 frames in the N dimension; the data operand is ONE 16-byte global load per lane and channel
// (lane (n, kg): frames t0 + o + 16 n + 8 kg ..+8), the chunks 1 and 2 are DPP row shifts of it; the Toeplitz operand is read from an
// LDS "sliding window" image (8 bytes per window start).  Checks the result against a host FIR and times the loop.
// Build: hipcc --offload-arch=gfx950 -O3 tools/diag/probe_t16.hip -o tools/diag/probe_t16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#define LDSP __attribute__((address_space(3)))

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) { bf16x2 v = {(__bf16)lo, (__bf16)hi}; return __builtin_bit_cast(unsigned, v); }

template <int SH>
__device__ __forceinline__ u32x4 row_shl(u32x4 v) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v[i], 0x100 + SH, 0xf, 0xf, true);
  return r;
}

// x: bf16 [C][pitch]; img: [C/16][16][CH] bytes; y: bf16 [C][192]
template <int NC, bool CHECK>
__global__ __launch_bounds__(256) void probe(const unsigned short* x, const unsigned char* img, unsigned short* y, int pitch, int t0o,
                                             int n_stage, long long* cyc) {
  constexpr int CH = (32 * NC + 16) * 8;
  constexpr int IMG = 16 * CH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const timg = smem + wave * IMG;
  char* const dwt = smem + 4 * IMG + wave * (16 * 512);
  const int n = lane & 15, kg = lane >> 4;
  const int s0 = 8 * kg - n + 15;          // A-operand lane map: row m = lane & 15
  const unsigned long long p = (unsigned long long)img;
  const i32x4 rsrc = {(int)(p & 0xffffffffu), (int)((p >> 32) & 0xffff), 0x7fffffff, 0x00020000};
  const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(LDSP char*)timg);
  long long t_begin = 0;
  for (int s = 0; s < n_stage; ++s) {
    const int grp = (blockIdx.x * n_stage + s) * 4 + wave;     // 16-channel group
    // image of this group: DMA, 1 KiB per instruction
#pragma unroll
    for (int h = 0; h < (IMG + 1023) / 1024; ++h)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(ldsb + h * 1024), "v"(lane * 16), "s"(rsrc), "s"(grp * IMG + h * 1024) : "memory");
    const unsigned short* xr = x + (size_t)grp * 16 * pitch + t0o + 16 * n + 8 * kg;
    u32x4 X[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) X[c] = *reinterpret_cast<const u32x4*>(xr + (size_t)c * pitch);
    __builtin_amdgcn_s_waitcnt(0x0070);    // vmcnt(0)
    asm volatile("s_nop 7" ::: "memory");
    if (s == 1) t_begin = clock64();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const u32x4 b0 = X[c];
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        const u32x4 b = cc == 0 ? b0 : (cc == 1 ? row_shl<2>(b0) : row_shl<4>(b0));
        const char* tp = timg + c * CH + 8 * s0 + cc * 256;
        const u32x2 a0 = *reinterpret_cast<const u32x2*>(tp), a1 = *reinterpret_cast<const u32x2*>(tp + 32);
        const u32x4 a = {a0[0], a0[1], a1[0], a1[1]};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), acc, 0, 0, 0);
      }
      *reinterpret_cast<u32x2*>(dwt + c * 512 + (16 * n + 4 * kg) * 2) = u32x2{pack_bf16(acc[0], acc[1]), pack_bf16(acc[2], acc[3])};
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0)
    if (CHECK) {
      for (int c = 0; c < 16; ++c)
        for (int t = lane; t < 192; t += 64) y[(size_t)(grp * 16 + c) * 192 + t] = *reinterpret_cast<unsigned short*>(dwt + c * 512 + t * 2);
    }
  }
  if (lane == 0 && wave == 0 && blockIdx.x == 0) cyc[0] = clock64() - t_begin;
}


// ---- pipelined form: what the producer wave of the fused kernel runs.  All vector-memory operations are inline asm (issued in
// program order, invisible to hipcc's own vmcnt model), waits are counted by hand.
template <int N>
__device__ __forceinline__ void vm_wait_x(u32x4& x) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(N) : "memory");
}
__device__ __forceinline__ u32x4 load16(i32x4 rsrc, int voff, int soff) {
  u32x4 r;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  return r;
}
template <int NC>
__global__ __launch_bounds__(256) void probe2(const unsigned short* x, const unsigned char* img, int pitch, int t0o, int n_stage, long long* cyc) {
  constexpr int CH = (32 * NC + 16) * 8;
  constexpr int IMG = 16 * CH;
  constexpr int D = IMG / 2048;                // DMAs per image half
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const timg = smem + wave * IMG;
  char* const dwt = smem + 4 * IMG + wave * (16 * 512);
  const int n = lane & 15, kg = lane >> 4;
  const int s0 = 8 * kg - n + 15;
  const unsigned long long p = (unsigned long long)img, px = (unsigned long long)x;
  const i32x4 rsrc = {(int)(p & 0xffffffffu), (int)((p >> 32) & 0xffff), 0x7fffffff, 0x00020000};
  const i32x4 rx = {(int)(px & 0xffffffffu), (int)((px >> 32) & 0xffff), 0x7fffffff, 0x00020000};
  const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(LDSP char*)timg);
  const int lane_x = (t0o + 16 * n + 8 * kg) * 2;
  const int pitch2 = pitch * 2;
  auto dma = [&](int half, int grp, float after) {
#pragma unroll
    for (int h = 0; h < D; ++h)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                   :: "s"(ldsb + (half * D + h) * 1024), "v"(lane * 16), "s"(rsrc), "s"(grp * IMG + (half * D + h) * 1024), "v"(after) : "memory");
  };
  u32x4 X[16];
  const int bx = n_stage < 0 ? blockIdx.x : (blockIdx.x & 1);   // L2-resident working set: two data sets
  int grp = (bx * n_stage) * 4 + wave;
#pragma unroll
  for (int c = 0; c < 16; ++c) X[c] = load16(rx, lane_x, (grp * 16 + c) * pitch2);
  dma(0, grp, 0.f); dma(1, grp, 0.f);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t_begin = 0;
  for (int s = 0; s < n_stage; ++s) {
    const int grp_next = (bx * n_stage + (s + 1 < n_stage ? s + 1 : s)) * 4 + wave;
    if (s == 1) t_begin = clock64();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c == 0 || c == 8) vm_wait_x<8 + D>(X[c]); else vm_wait_x<15 + 2 * D>(X[c]);
      const u32x4 b0 = X[c];
      u32x4 a[NC];
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        const char* tp = timg + c * CH + 8 * s0 + cc * 256;
        const u32x2 a0 = *reinterpret_cast<const u32x2*>(tp), a1 = *reinterpret_cast<const u32x2*>(tp + 32);
        a[cc] = u32x4{a0[0], a0[1], a1[0], a1[1]};
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        const u32x4 b = cc == 0 ? b0 : (cc == 1 ? row_shl<2>(b0) : row_shl<4>(b0));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16x8, a[cc]), __builtin_bit_cast(s16x8, b), acc, 0, 0, 0);
      }
      X[c] = load16(rx, lane_x, (grp_next * 16 + c) * pitch2);
      *reinterpret_cast<u32x2*>(dwt + c * 512 + (16 * n + 4 * kg) * 2) = u32x2{pack_bf16(acc[0], acc[1]), pack_bf16(acc[2], acc[3])};
      if (c == 7) dma(0, grp_next, acc[0]);
      if (c == 15) dma(1, grp_next, acc[0]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (lane == 0 && wave == 0 && blockIdx.x == 0) cyc[0] = clock64() - t_begin;
}

template <int NC>
int run(int K) {
  const int pad = K / 2, o = -8 * ((pad + 7) / 8);
  if (o + 32 * NC < 16 + (K - 1 - pad)) { printf("K %d does not fit NC %d\n", K, NC); return 1; }
  const int U0 = o + pad;
  constexpr int CH = (32 * NC + 16) * 8, IMG = 16 * CH;
  const int n_stage = 8, n_wg = 256, C = n_wg * n_stage * 64, pitch = 1152, T = 751, t0 = 192;
  std::vector<unsigned short> hx((size_t)C * pitch + 2048, 0), hw((size_t)C * K);
  srand(1);
  for (int c = 0; c < C; ++c) {
    for (int t = 0; t < T; ++t) hx[1024 + (size_t)c * pitch + t] = f2bf((rand() % 2001 - 1000) * 1e-3f);
    for (int u = 0; u < K; ++u) hw[(size_t)c * K + u] = f2bf((rand() % 2001 - 1000) * 1e-3f);
  }
  std::vector<unsigned char> himg((size_t)C * CH, 0);
  for (int c = 0; c < C; ++c)
    for (int pp = 0; pp < 32 * NC + 16; ++pp)
      for (int j = 0; j < 4; ++j) {
        const int u = U0 + pp + j - 15;
        const unsigned short v = (u >= 0 && u < K) ? hw[(size_t)c * K + u] : 0;
        memcpy(&himg[(size_t)c * CH + pp * 8 + j * 2], &v, 2);
      }
  unsigned short *dx, *dy; unsigned char* di; long long* dc;
  hipMalloc(&dx, hx.size() * 2); hipMalloc(&dy, (size_t)C * 192 * 2); hipMalloc(&di, himg.size()); hipMalloc(&dc, 8);
  hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(di, himg.data(), himg.size(), hipMemcpyHostToDevice);
  const int lds = 4 * IMG + 4 * 16 * 512;
  hipFuncSetAttribute((const void*)probe<NC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)probe<NC, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((probe<NC, true>), dim3(n_wg), dim3(256), lds, 0, dx + 1024, di, dy, pitch, t0 + o, n_stage, dc);
  std::vector<unsigned short> hy((size_t)C * 192);
  hipMemcpy(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost);
  double maxerr = 0; int bad = 0;
  for (int c = 0; c < C; c += 7)
    for (int t = 0; t < 192; ++t) {
      double ref = 0;
      for (int u = 0; u < K; ++u) {
        const int ti = t0 + t + u - pad;
        if (ti >= 0 && ti < T) ref += (double)bf2f(hw[(size_t)c * K + u]) * bf2f(hx[1024 + (size_t)c * pitch + ti]);
      }
      const double got = bf2f(hy[(size_t)c * 192 + t]);
      const double e = fabs(got - ref);
      if (e > maxerr) maxerr = e;
      if (e > 0.02 + 0.01 * fabs(ref)) { if (bad < 5) printf("  mismatch c %d t %d got %f ref %f\n", c, t, got, ref); ++bad; }
    }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<NC, false>), dim3(n_wg), dim3(256), lds, 0, dx + 1024, di, dy, pitch, t0 + o, n_stage, dc);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<NC, false>), dim3(n_wg), dim3(256), lds, 0, dx + 1024, di, dy, pitch, t0 + o, n_stage, dc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long cyc; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
  hipFuncSetAttribute((const void*)probe2<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe2<NC>), dim3(n_wg), dim3(256), lds, 0, dx + 1024, di, pitch, t0 + o, n_stage, dc);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe2<NC>), dim3(n_wg), dim3(256), lds, 0, dx + 1024, di, pitch, t0 + o, n_stage, dc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms2; hipEventElapsedTime(&ms2, e0, e1);
  long long cyc2; hipMemcpy(&cyc2, dc, 8, hipMemcpyDeviceToHost);
  printf("   pipelined: %.1f us per launch (%.2f us/stage), in-kernel %lld cycles per stage\n", ms2 * 1e3 / 20, ms2 * 1e3 / 20 / n_stage, cyc2 / (n_stage - 1));
  printf("K %2d NC %d o %3d: max err %.4f, %d mismatches;  %.1f us per launch of %d stages (%.2f us/stage), in-kernel %lld cycles for %d stages = %lld per stage\n",
         K, NC, o, maxerr, bad, ms * 1e3 / 20, n_stage, ms * 1e3 / 20 / n_stage, cyc, n_stage - 1, cyc / (n_stage - 1));
  hipFree(dx); hipFree(dy); hipFree(di); hipFree(dc);
  return bad != 0;
}

int main() {
  int rc = 0;
  rc |= run<1>(11); rc |= run<1>(17);
  rc |= run<2>(33); rc |= run<2>(39);
  rc |= run<3>(51); rc |= run<3>(63); rc |= run<3>(75);
  return rc;
}
