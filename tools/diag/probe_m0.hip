// Probe: N back-to-back `buffer_load_dwordx4 ... lds` with a different M0 each (slots 1 KiB apart), GAP wait states
// between a DMA and the next M0 write.  Every lane of every slot is checked after a full drain.
// Build: hipcc --offload-arch=gfx950 -O3 tools/diag/probe_m0.hip -o tools/diag/probe_m0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define LDSP __attribute__((address_space(3)))

template <int NDMA, int GAP>
__global__ __launch_bounds__(512) void probe(const unsigned* src, unsigned* bad, unsigned* hist, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* region = smem + wave * (NDMA * 1024);
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(LDSP char*)region);
  const unsigned long long p = (unsigned long long)src;
  const i32x4 rsrc = {(int)(p & 0xffffffffu), (int)((p >> 32) & 0xffff), 0x7fffffff, 0x00020000};
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NDMA; ++j) *(u32x4*)(region + j * 1024 + lane * 16) = u32x4{0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
    __builtin_amdgcn_s_waitcnt(0x0070);
    const int off = __builtin_amdgcn_readfirstlane(((it * 61 + blockIdx.x * 8 + wave) & 4095) * 8192);
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 1\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(base + j * 1024), "v"(lane * 16), "s"(rsrc), "s"(off + j * 1024) : "memory");
      if (GAP >= 8) { for (int g = 0; g < GAP / 8; ++g) asm volatile("s_nop 7" ::: "memory"); }
      else if (GAP > 0) asm volatile("s_nop %0" :: "n"(GAP > 0 ? GAP - 1 : 0) : "memory");
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
      const u32x4 g = *(u32x4*)(region + j * 1024 + lane * 16);
      const unsigned e = src[(off + j * 1024 + lane * 16) / 4];
      if (g[0] != e) { ++nbad; atomicAdd(&hist[j * 4 + (lane >> 4)], 1u); }
    }
  }
  if (nbad) atomicAdd(bad, nbad);
}

template <int NDMA, int GAP>
void run(const unsigned* src, unsigned* bad, unsigned* hist) {
  hipMemset(bad, 0, 4); hipMemset(hist, 0, 64 * 4);
  hipFuncSetAttribute((const void*)probe<NDMA, GAP>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * NDMA * 1024);
  hipLaunchKernelGGL((probe<NDMA, GAP>), dim3(256), dim3(512), 8 * NDMA * 1024, 0, src, bad, hist, 200);
  unsigned h = 0, hh[64];
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(hh, hist, 64 * 4, hipMemcpyDeviceToHost);
  printf("%d DMAs back to back, %3d wait states before the next M0 write: %u bad lane-slots of %d;  by (dma, 16-lane group):", NDMA, GAP, h, 256 * 512 * 200 * NDMA);
  for (int j = 0; j < NDMA * 4; ++j) if (hh[j]) printf(" (%d,%d)=%u", j / 4, j % 4, hh[j]);
  printf("\n");
}

int main() {
  const size_t n = 64u << 20;
  std::vector<unsigned> h(n / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u) | 1u;
  unsigned *src, *bad, *hist;
  hipMalloc(&src, n); hipMalloc(&bad, 4); hipMalloc(&hist, 64 * 4);
  hipMemcpy(src, h.data(), n, hipMemcpyHostToDevice);
  run<2, 0>(src, bad, hist); run<4, 0>(src, bad, hist); run<6, 0>(src, bad, hist); run<8, 0>(src, bad, hist);
  run<6, 8>(src, bad, hist); run<6, 32>(src, bad, hist); run<6, 128>(src, bad, hist); run<8, 128>(src, bad, hist);
  return 0;
}
