// Fine-tuning with a frozen encoder (the first phase of the reference's FinetuneCTCModule recipe: callbacks.py freezes
// the encoder until `unfreeze_encoder_at_epoch`): the trainable part of the step is the 1x1 decoder.
//   ts_decoder_bwd : dW[v, c] = sum_{b,t} g[b, v, t] * x[b, c, t],  db[v] = sum_{b,t} g[b, v, t]
//                    (backward of conv1d_decoder, reference blocks.py:199-216; g = dL/dlogits from ts_ctc_loss)
//   ts_adamw_step  : torch.optim.AdamW update (decoupled weight decay), one launch per flat parameter buffer
#include "ts_common.hpp"

namespace ts {

constexpr int WG_C = 64;      // input channels per workgroup
constexpr int WG_T = 128;     // frames per LDS tile
constexpr int WG_V = 32;      // classes per pass

// grid: (ceil(C / 64), ceil(B / clips_per_wg)); 256 threads: thread = (c = tid & 63, vg = tid >> 6 -> classes vg*8 .. +8)
__global__ __launch_bounds__(256) void decoder_wgrad_kernel(const float* __restrict__ g, const unsigned short* __restrict__ x,
                                                            float* __restrict__ dw, float* __restrict__ db, int batch, int n_cls,
                                                            int channels, int t, int pitch_g, int pitch_x, int clips_per_wg) {
  __shared__ float gs[WG_V][WG_T + 1];
  __shared__ float xs[WG_C][WG_T + 1];
  const int tid = threadIdx.x;
  const int c_l = tid & 63, vg = tid >> 6;
  const int c0 = blockIdx.x * WG_C;
  const int b0 = blockIdx.y * clips_per_wg;
  const int b1 = b0 + clips_per_wg < batch ? b0 + clips_per_wg : batch;
  for (int v0 = 0; v0 < n_cls; v0 += WG_V) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    float bsum = 0.f;
    for (int b = b0; b < b1; ++b) {
      for (int t0 = 0; t0 < t; t0 += WG_T) {
        __syncthreads();
        // stage g[b, v0 .. v0+32, t0 .. t0+128) and x[b, c0 .. c0+64, t0 .. t0+128): time-contiguous, coalesced
        for (int i = tid; i < WG_V * WG_T; i += 256) {
          const int v = i / WG_T, tt = i % WG_T;
          gs[v][tt] = (v0 + v < n_cls && t0 + tt < t) ? g[((size_t)b * n_cls + v0 + v) * pitch_g + t0 + tt] : 0.f;
        }
        for (int i = tid; i < WG_C * WG_T; i += 256) {
          const int c = i / WG_T, tt = i % WG_T;
          xs[c][tt] = (c0 + c < channels && t0 + tt < t) ? bf16_to_f32(x[((size_t)b * channels + c0 + c) * pitch_x + t0 + tt]) : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int tt = 0; tt < WG_T; ++tt) {
          const float xv = xs[c_l][tt];
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += gs[vg * 8 + i][tt] * xv;
        }
        if (blockIdx.x == 0 && tid < WG_V) {
          float s = 0.f;
          for (int tt = 0; tt < WG_T; ++tt) s += gs[tid][tt];
          bsum += s;
        }
      }
    }
    if (c0 + c_l < channels) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int v = v0 + vg * 8 + i;
        if (v < n_cls) atomicAdd(dw + (size_t)v * channels + c0 + c_l, acc[i]);
      }
    }
    if (blockIdx.x == 0 && tid < WG_V && v0 + tid < n_cls) atomicAdd(db + v0 + tid, bsum);
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float lr, float beta1, float beta2,
                                                    float eps, float weight_decay, float bias_c1, float bias_c2) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  float pi = p[i] * (1.f - lr * weight_decay);
  const float mi = beta1 * m[i] + (1.f - beta1) * gi;
  const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  pi -= (lr / bias_c1) * mi / (sqrtf(vi) / sqrtf(bias_c2) + eps);
  p[i] = pi;
}

// all tensors of a parameter group in ONE launch: table[i] = (param, grad, exp_avg, exp_avg_sq, n, bf16 shadow | 0) as 64-bit words;
// blockIdx.y is the tensor, blockIdx.x its 1024-element block (blocks past a tensor's end leave at once).  A non-zero shadow pointer
// receives the updated parameter rounded to bf16: the GEMM operand copy of the mixed-precision training path, refreshed for free.
__global__ __launch_bounds__(256) void adamw_multi_kernel(const unsigned long long* __restrict__ table, float lr, float beta1, float beta2,
                                                          float eps, float weight_decay, float bias_c1, float bias_c2) {
  const unsigned long long* e = table + (size_t)blockIdx.y * 6;
  const long long n = (long long)e[4];
  const long long i0 = (long long)blockIdx.x * 1024 + threadIdx.x;
  if ((long long)blockIdx.x * 1024 >= n) return;
  float* p = reinterpret_cast<float*>(e[0]);
  const float* g = reinterpret_cast<const float*>(e[1]);
  float* m = reinterpret_cast<float*>(e[2]);
  float* v = reinterpret_cast<float*>(e[3]);
  unsigned short* shadow = reinterpret_cast<unsigned short*>(e[5]);
  const float step_size = lr / bias_c1, inv_sqrt_c2 = 1.f / sqrtf(bias_c2), decay = 1.f - lr * weight_decay;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long i = i0 + 256 * j;
    if (i < n) {
      const float gi = g[i];
      float pi = p[i] * decay;
      const float mi = beta1 * m[i] + (1.f - beta1) * gi;
      const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
      m[i] = mi;
      v[i] = vi;
      pi -= step_size * mi / (sqrtf(vi) * inv_sqrt_c2 + eps);
      p[i] = pi;
      if (shadow) shadow[i] = (unsigned short)(pack_bf16(pi, 0.f) & 0xffffu);
    }
  }
}

// Gradient wire format of the data-parallel exchange (bf16 buckets over xGMI): pack = scale (1 / world) and round
// to bf16, unpack = widen back into the fp32 gradient buffer.  8 elements per thread, 16-byte stores.
__global__ __launch_bounds__(256) void grad_pack_kernel(const float* __restrict__ g, unsigned short* __restrict__ w, long long n, float scale) {
  const long long n8 = n >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(g + 8 * i), b = *reinterpret_cast<const f32x4*>(g + 8 * i + 4);
    *reinterpret_cast<u32x4*>(w + 8 * i) = u32x4{pack_bf16(a[0] * scale, a[1] * scale), pack_bf16(a[2] * scale, a[3] * scale),
                                                  pack_bf16(b[0] * scale, b[1] * scale), pack_bf16(b[2] * scale, b[3] * scale)};
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const long long i = (n8 << 3) + threadIdx.x;
    w[i] = (unsigned short)(pack_bf16(g[i] * scale, 0.f) & 0xffffu);
  }
}
__global__ __launch_bounds__(256) void grad_unpack_kernel(const unsigned short* __restrict__ w, float* __restrict__ g, long long n) {
  const long long n8 = n >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(w + 8 * i);
    *reinterpret_cast<f32x4*>(g + 8 * i) = f32x4{bf16_lo(v[0]), bf16_hi(v[0]), bf16_lo(v[1]), bf16_hi(v[1])};
    *reinterpret_cast<f32x4*>(g + 8 * i + 4) = f32x4{bf16_lo(v[2]), bf16_hi(v[2]), bf16_lo(v[3]), bf16_hi(v[3])};
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const long long i = (n8 << 3) + threadIdx.x;
    g[i] = bf16_to_f32(w[i]);
  }
}

}  // namespace ts

extern "C" int ts_grad_wire_pack(const float* grad, void* wire_bf16, int64_t n, float scale, void* stream) {
  if (!grad || !wire_bf16 || n <= 0) return TS_EINVAL;
  if ((reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(wire_bf16)) & 15) return TS_EINVAL;
  const long long blocks = ((n >> 3) + 255) / 256;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::grad_pack_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), grad, static_cast<unsigned short*>(wire_bf16), (long long)n, scale);
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_grad_wire_unpack(const void* wire_bf16, float* grad, int64_t n, void* stream) {
  if (!grad || !wire_bf16 || n <= 0) return TS_EINVAL;
  if ((reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(wire_bf16)) & 15) return TS_EINVAL;
  const long long blocks = ((n >> 3) + 255) / 256;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::grad_unpack_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), static_cast<const unsigned short*>(wire_bf16), grad, (long long)n);
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_adamw_multi_step(const void* table, int32_t n_tensors, int64_t max_numel, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, int32_t step, void* stream_) {
  if (!table || n_tensors <= 0 || max_numel <= 0 || step <= 0) return TS_EINVAL;
  if (n_tensors > 65535) return TS_EUNSUPPORTED;
  const float c1 = 1.f - powf(beta1, (float)step), c2 = 1.f - powf(beta2, (float)step);
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::adamw_multi_kernel, dim3((unsigned)((max_numel + 1023) / 1024), n_tensors), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream_), static_cast<const unsigned long long*>(table), lr, beta1, beta2, eps,
                     weight_decay, c1, c2);
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_decoder_bwd(const float* grad_logits, const void* x, int32_t batch, int32_t n_classes, int32_t channels,
                              int32_t t, int32_t pitch_g, int32_t pitch_x, float* d_weight, float* d_bias, void* stream_) {
  if (!grad_logits || !x || !d_weight || !d_bias) return TS_EINVAL;
  if (batch <= 0 || n_classes <= 0 || channels <= 0 || t <= 0 || pitch_g < t || pitch_x < t) return TS_EINVAL;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipError_t e = hipMemsetAsync(d_weight, 0, sizeof(float) * (size_t)n_classes * channels, stream);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(d_bias, 0, sizeof(float) * (size_t)n_classes, stream);
  if (e != hipSuccess) return (int)e;
  const int clips_per_wg = 1;      // 16 channel tiles x B clips: enough workgroups to fill the chip (4 clips per workgroup left half of it idle)
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::decoder_wgrad_kernel, dim3((channels + ts::WG_C - 1) / ts::WG_C, (batch + clips_per_wg - 1) / clips_per_wg),
                     dim3(256), 0, stream, grad_logits, static_cast<const unsigned short*>(x), d_weight, d_bias, batch, n_classes,
                     channels, t, pitch_g, pitch_x, clips_per_wg);
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int32_t step, void* stream_) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return TS_EINVAL;
  const float c1 = 1.f - powf(beta1, (float)step), c2 = 1.f - powf(beta2, (float)step);
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), param,
                     grad, exp_avg, exp_avg_sq, (long long)n, lr, beta1, beta2, eps, weight_decay, c1, c2);
  return ts::hip_status(hipGetLastError());
}
