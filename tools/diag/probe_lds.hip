// Probe: LDS read throughput per CU for the three read flavours the TCS kernels use, 8 waves per workgroup, 1 WG per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/diag/probe_lds.hip -o tools/diag/probe_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define LDSP __attribute__((address_space(3)))

template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<unsigned*>(smem)[i] = i;
  __syncthreads();
  unsigned acc = 0;
  // MODE 0: ds_read_b64_tr_b16 (rows of 256 B, the pattern of the A-fragment reads); 1: ds_read_b64 contiguous; 2: ds_read_b128
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (MODE == 0) {
        const int q4 = (lane >> 2) & 3, p4 = lane & 3, h = lane >> 5, gq = (lane >> 4) & 1;
        const int c = 8 * h + q4, t = 16 * gq + 4 * p4 + 32 * (j & 3);
        const int off = ((wave * 4 + (j >> 2)) & 3) * 16384 + c * 256 + ((((t >> 3) ^ ((c & 3) * 5))) << 4) + ((t & 7) << 1);
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP s16x4*)((LDSP char*)smem + off));
        acc += (unsigned)v[0] + (unsigned)v[3];
      } else if (MODE == 1) {
        u32x2 v = *(u32x2*)(smem + ((wave * 16 + j) & 63) * 1024 + lane * 8 + ((j & 1) ? 512 : 0));
        acc += v[0] + v[1];
      } else {
        u32x4 v = *(u32x4*)(smem + ((wave * 16 + j) & 63) * 1024 + lane * 16);
        acc += v[0] + v[3];
      }
    }
  }
  sink[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int MODE>
void run(unsigned* sink, const char* name, int bytes_per_lane) {
  hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(512), 65536, 0, sink, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(512), 65536, 0, sink, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 512.0 * bytes_per_lane * 16 * iters;        // per CU
  printf("%-22s %8.3f ms  %7.1f GB/s per CU  (%5.1f B/clk at 2.1 GHz)\n", name, ms, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 2.1e9);
}

int main() {
  unsigned* sink; hipMalloc(&sink, 256 * 512 * 4);
  run<0>(sink, "ds_read_b64_tr_b16", 8);
  run<1>(sink, "ds_read_b64", 8);
  run<2>(sink, "ds_read_b128", 16);
  return 0;
}
