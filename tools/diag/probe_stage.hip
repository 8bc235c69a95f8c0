// Probe: the STAGE LOOP of the fused TCS kernel without global memory -- LDS traffic, matrix-core work and one s_barrier per stage,
// for the wave structures under discussion.  What it answers: how long does one 64-channel stage take
// when its instruction mix is spread over 12 waves (the shipped split kernel: 8 consumers of 96 x 64 + 4 producers of 16 channels
// x 96 frames) or over 16 waves at 128 VGPRs (8 consumers of 64 x 64 + 8 producers, each 16 channels x 32 frames, or 8 channels x
// 64 frames as 2 x 8 blocks)?  Every wave executes the real kernel's per-stage instruction counts on LDS images with the real
// kernel's bank geometry; operands are random bits (the matrix core's clock depends on the data).
//   hipcc --offload-arch=gfx950 -O3 -o probe_stage probe_stage.hip && ./probe_stage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#define LDS __attribute__((address_space(3)))
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
constexpr int STAGES = 2048;

// NC consumer waves (MT x 2 accumulators of 32 x 32), NPW producer waves (M steps of 4 frames per lane run, NK k-steps of taps,
// XW staged 8-byte row writes per lane), KS k-steps of 16 channels per stage in the consumers (4 = 64 channels)
// OFF bits (experiments): 1 producers issue no MFMA, 2 producers touch no LDS, 4 consumers read no LDS, 8 no barrier, 16 producers are the
// workgroup's oldest waves, 32 producers run at s_setprio 3
template <int NC, int NPW, int MT, int M, int NK, int XW, bool CONS, bool PROD, bool C16 = false, int OFF = 0>
__global__ __launch_bounds__((NC + NPW) * 64) void k_stage(float* o, const unsigned* seed, long long* cyc) {
  constexpr int TT = 32 * MT;
  constexpr int ROWB = 256;                      // dw tile row pitch (bytes), >= 2 TT
  constexpr int TILEB = 64 * ROWB;
  constexpr int NP = NK + M - 1;
  constexpr int CST = (16 * NK + 16) % 32 == 16 ? 16 * NK + 16 : 16 * NK + 32;
  constexpr int XPITCH = (TT + 4 * NK + 8 + 63) / 64 * 64 + 4;   // staged row pitch (elements): == 8 mod 16 bytes
  constexpr int XSB = 16 * XPITCH * 2;
  constexpr int TAPB = (16 * CST + 1023) / 1024 * 1024;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const dwt = smem;                                   // [2][64][ROWB]
  char* const prod0 = smem + 2 * TILEB;                     // [NPW][XSB + TAPB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_hw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = (OFF & 16) ? (wave_hw < NPW ? NC + wave_hw : wave_hw - NPW) : wave_hw;      // bit 4: the producers are the oldest waves
  for (int i = tid; i < (2 * TILEB + NPW * (XSB + TAPB)) / 4; i += blockDim.x)
    reinterpret_cast<unsigned*>(smem)[i] = seed[i & 4095] & 0x3f803f80u;      // small finite bf16 pairs
  __syncthreads();
  auto taddr = [](int c, int t) { return c * ROWB + ((((t >> 3) ^ ((c & 3) * 5))) << 4) + ((t & 7) << 1); };
  float r = 0.f;
  long long t0 = 0, t1 = 0;
  if (wave < NC) {
    const int h = lane >> 5, gq = (lane >> 4) & 1, q4 = (lane >> 2) & 3, p4 = lane & 3;
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) abase[mt] = taddr(8 * h + q4, 32 * mt + 16 * gq + 4 * p4);
    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    // C16: the same output tile on v_mfma_f32_16x16x32_bf16: 2 MT x 4 accumulators of 16 x 16, k-steps of 32 channels
    f32x4 acc16[C16 ? 2 * MT : 1][4];
    int abase16[C16 ? 2 * MT : 1];
    if (C16) {
#pragma unroll
      for (int i = 0; i < 2 * MT; ++i) {
        abase16[i] = taddr(8 * (lane >> 4) + q4, 16 * i + 4 * p4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    s16x8 bw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bw[j] = *reinterpret_cast<const s16x8*>(seed + 256 * j + lane * 4);
    __syncthreads();
    t0 = clock64();
    for (int s = 0; s < STAGES; ++s) {
      const char* src = dwt + (s & 1) * TILEB;
      if (CONS && C16) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s16x8 af[2 * MT];
#pragma unroll
          for (int mt = 0; mt < 2 * MT; ++mt) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS s16x4*)((LDS char*)src + abase16[mt] + ks * 32 * ROWB));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS s16x4*)((LDS char*)src + abase16[mt] + ks * 32 * ROWB + 4 * ROWB));
            af[mt] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2 * MT; ++mt) acc16[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bw[nt], acc16[mt][nt], 0, 0, 0);
        }
      } else if (CONS) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s16x8 af[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if (OFF & 4) { af[mt] = bw[mt & 3]; continue; }
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS s16x4*)((LDS char*)src + abase[mt] + ks * 16 * ROWB));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS s16x4*)((LDS char*)src + abase[mt] + ks * 16 * ROWB + 4 * ROWB));
            af[mt] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bw[nt], acc[mt][nt], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(OFF & 8)) __builtin_amdgcn_s_barrier();
    }
    t1 = clock64();
#pragma unroll
    for (int i = 0; i < MT; ++i) r += acc[i][0][0] + acc[i][1][5];
    if (C16) for (int i = 0; i < 2 * MT; ++i) for (int j = 0; j < 4; ++j) r += acc16[i][j][0] + acc16[i][j][3];
  } else {
    const int pw = wave - NC;
    if (OFF & 32) __builtin_amdgcn_s_setprio(3);
    char* const xs = prod0 + (size_t)pw * (XSB + TAPB);
    char* const tapl = xs + XSB;
    const int row = lane >> 2, q = lane & 3;
    constexpr int RUN = TT / 4 >= 4 * M ? 4 * M : 4 * M;     // frames per lane run
    char* const xw = xs + ((size_t)row * XPITCH + q * 8) * 2;
    const char* const xrow = xs + ((size_t)row * XPITCH + q * RUN) * 2;
    const int tap_off = row * CST + ((lane & 1) ? 0 : 4) + ((lane & 3) < 2 ? 8 : 0);
    const int chan = (pw * 16 + row) & 63;
    int dw_out[M];
#pragma unroll
    for (int m = 0; m < M; ++m) dw_out[m] = taddr(chan, (q * RUN + 4 * m) % (ROWB / 2));
    f32x4 d[M];
    s16x4 P[NP];
    u32x2 T[NK];
    u32x2 xv = *reinterpret_cast<const u32x2*>(seed + lane * 2);
    __syncthreads();
    t0 = clock64();
    for (int s = 0; s < STAGES; ++s) {
      char* dst = dwt + ((s + 1) & 1) * TILEB;
      if (PROD) {
        const char* trow = tapl + tap_off;
        asm volatile("" : "+v"(trow));
        if (!(OFF & 2)) {
#pragma unroll
        for (int j = 0; j < XW; ++j) *reinterpret_cast<u32x2*>(xw + j * 64) = xv;      // staged rows (registers -> LDS)
        }
#pragma unroll
        for (int m = 0; m < M; ++m) d[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NP; ++u) P[u] = (OFF & 2) ? __builtin_bit_cast(s16x4, u32x2{xv[0] + u, xv[1]}) : *reinterpret_cast<const s16x4*>(xrow + u * 8);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
          T[kk] = (OFF & 2) ? u32x2{xv[1] + kk, xv[0]} : u32x2{*reinterpret_cast<const unsigned*>(trow + kk * 16), *reinterpret_cast<const unsigned*>(trow + kk * 16 + 8)};
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
#pragma unroll
          for (int m = 0; m < M; ++m) {
            if (OFF & 1) { d[m][0] += __builtin_bit_cast(float, T[kk][0] ^ (unsigned)P[kk + m][0]); continue; }
            d[m] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, T[kk]), P[kk + m], d[m], 0, 0, 0);
          }
#pragma unroll
        for (int m = 0; m < M; ++m) {
          typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
          bf2 a = {(__bf16)d[m][0], (__bf16)d[m][1]}, b = {(__bf16)d[m][2], (__bf16)d[m][3]};
          if (!(OFF & 2)) *reinterpret_cast<u32x2*>(dst + dw_out[m]) = u32x2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
          else xv[1] += __builtin_bit_cast(unsigned, a);
        }
        xv[0] ^= __builtin_bit_cast(unsigned, d[0][0]) & 0x00010001u;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(OFF & 8)) __builtin_amdgcn_s_barrier();
    }
    t1 = clock64();
    r = d[0][0] + (float)xv[0];
  }
  o[blockIdx.x * blockDim.x + tid] = r;
  if (blockIdx.x == 3 && lane == 0) cyc[wave] = t1 - t0;      // cyc[0]: a consumer wave
}

template <typename K> static int run(const char* name, K kern, int threads, size_t lds, int frames, float* o, unsigned* seed, long long* cyc) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), lds, 0, o, seed, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(256), dim3(threads), lds, 0, o, seed, cyc); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long c[16]; CK(hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost));
  const double ns = ms * 1e6 / STAGES;
  printf("%-58s %7.1f ns / stage of %3d frames = %6.1f ns per 96 frames; s_memtime ticks / stage (wave 0): %lld\n", name, ns, frames, ns * 96 / frames,
         c[0] / STAGES);
  return 0;
}

template <int NC, int NPW, int MT, int M, int NK, int XW>
static size_t lds_bytes() {
  constexpr int TT = 32 * MT, CST = (16 * NK + 16) % 32 == 16 ? 16 * NK + 16 : 16 * NK + 32;
  constexpr int XPITCH = (TT + 4 * NK + 8 + 63) / 64 * 64 + 4;
  return 2 * 64 * 256 + (size_t)NPW * (16 * XPITCH * 2 + (16 * CST + 1023) / 1024 * 1024);
}

int main() {
  float* o; unsigned* seed; long long* cyc;
  CK(hipMalloc(&o, 256 * 1024 * 4)); CK(hipMalloc(&seed, 4096 * 4)); CK(hipMalloc(&cyc, 16 * 8));
  unsigned h[4096]; srand(1); for (auto& v : h) v = (unsigned)rand() * 2654435761u;
  CK(hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice));
#define RUN16(label, NC, NPW, MT, M, NK, XW, C, P, frames) \
  if (run(label, k_stage<NC, NPW, MT, M, NK, XW, C, P, true>, (NC + NPW) * 64, lds_bytes<NC, NPW, MT, M, NK, XW>(), frames, o, seed, cyc)) return 1;
#define RUNX(label, OFF_) \
  if (run(label, k_stage<8, 4, 3, 6, 18, 12, true, true, false, OFF_>, 768, lds_bytes<8, 4, 3, 6, 18, 12>(), 96, o, seed, cyc)) return 1;
#define RUN(label, NC, NPW, MT, M, NK, XW, C, P, frames) \
  if (run(label, k_stage<NC, NPW, MT, M, NK, XW, C, P>, (NC + NPW) * 64, lds_bytes<NC, NPW, MT, M, NK, XW>(), frames, o, seed, cyc)) return 1;
  // K63 (NK = 18) and K33 (NK = 9)
  RUN("12 waves: 8C(96x64) + 4P(16ch x 96fr)  K63  both", 8, 4, 3, 6, 18, 12, true, true, 96)
  RUN("12 waves:                                     consumers only", 8, 4, 3, 6, 18, 12, true, false, 96)
  RUN("12 waves:                                     producers only", 8, 4, 3, 6, 18, 12, false, true, 96)
  RUNX("  producers = the OLDEST waves (0..3)", 16)
  RUNX("  producers at s_setprio 3", 32)
  RUNX("  producers oldest AND s_setprio 3", 48)
  RUNX("  switch-off: producers issue no MFMA", 1)
  RUNX("  switch-off: producers touch no LDS", 2)
  RUNX("  switch-off: consumers read no LDS", 4)
  RUNX("  switch-off: no barrier (roles free-running)", 8)
  RUNX("  switch-off: producers no LDS + consumers no LDS", 6)
  RUNX("  switch-off: producers no MFMA, no LDS (VALU only)", 3)
  RUN16("12 waves, consumers on 16x16x32:              K63  both", 8, 4, 3, 6, 18, 12, true, true, 96)
  RUN16("12 waves, consumers on 16x16x32:              consumers only", 8, 4, 3, 6, 18, 12, true, false, 96)
  RUN("16 waves: 8C(64x64) + 8P(16ch x 32fr)  K63  both", 8, 8, 2, 2, 18, 4, true, true, 64)
  RUN("16 waves:                                     consumers only", 8, 8, 2, 2, 18, 4, true, false, 64)
  RUN("16 waves:                                     producers only", 8, 8, 2, 2, 18, 4, false, true, 64)
  RUN("12 waves: 4C(96x64) + 8P(8ch x 96fr as 2x48) K33  both", 4, 8, 3, 3, 9, 6, true, true, 96)
  RUN("12 waves:                                     consumers only", 4, 8, 3, 3, 9, 6, true, false, 96)
  RUN("12 waves:                                     producers only", 4, 8, 3, 3, 9, 6, false, true, 96)
  RUN("12 waves: 8C + 4P(16ch x 96fr)  K33  both  [c_out 512]", 8, 4, 3, 6, 9, 12, true, true, 96)
  return 0;
}
