// What does one workgroup barrier round cost on gfx950?  64 workgroups x NT threads loop N times over
//   mode 0: s_barrier only;  1: ds_write + lgkmcnt(0) + s_barrier + ds_read + lgkmcnt(0);  2: mode 1 + a 12-op dependent VALU chain (2 exp, 1 log)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_barrier tools/diag/probe_barrier.hip && /tmp/probe_barrier
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(float* out, int n) {
  __shared__ float buf[2][1024 + 8];
  const int lane = threadIdx.x;
  float v = lane * 1e-3f;
  buf[0][lane + 2] = v; buf[1][lane + 2] = v;
  __syncthreads();
  for (int i = 0; i < n; ++i) {
    if (MODE >= 1) {
      float* cur = buf[i & 1];
      float* prev = buf[(i & 1) ^ 1];
      float a = prev[lane + 1], b = prev[lane];
      if (MODE >= 2) {
        const float m = fmaxf(fmaxf(v, a), b);
        const float s = __builtin_amdgcn_exp2f((v - m) * 1.44f) + __builtin_amdgcn_exp2f((a - m) * 1.44f) + __builtin_amdgcn_exp2f((b - m) * 1.44f);
        v = m + __builtin_amdgcn_logf(s) * 0.69f - 1.0f;
      } else {
        v = v + a + b;
      }
      cur[lane + 2] = v;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  out[blockIdx.x * blockDim.x + lane] = v;
}

template <int MODE>
void run(int nt, int n, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(64), dim3(nt), 0, 0, out, 1000);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(64), dim3(nt), 0, 0, out, n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("mode %d threads %4d: %.1f ns per iteration\n", MODE, nt, ms * 1e6 / n);
}

int main() {
  float* out; hipMalloc(&out, 64 * 1024 * 4);
  for (int nt : {64, 128, 256, 512, 768, 1024}) {
    run<0>(nt, 200000, out); run<1>(nt, 200000, out); run<2>(nt, 200000, out);
  }
  return 0;
}
