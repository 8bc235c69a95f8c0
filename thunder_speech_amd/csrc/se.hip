// Squeeze-excite of the Citrinet blocks (reference citrinet/blocks.py:70-83, applied at :154, :186-196):
//   g[b, c]   = sigmoid( W2 . relu( W1 . mean_t y[b, :, t] ) )        mean over ALL T frames, padded ones included (quirk A3)
//   out[b,c,t] = relu( g[b, c] * y[b, c, t] + r[b, c, t] )            r = BN(conv1x1(x_block)) or absent
// y and r come out of the fused sub-block launches with their tails zeroed (tail-zero invariant); the reference's
// values beyond the length are constants -- a masked input makes conv output 0, so y = r = the folded BN shift there --
// and are re-inserted analytically (`tail_y`, `tail_r`), both in the pooled mean and in the caller-visible tail.
#include "ts_common.hpp"

namespace ts {

// one wave per (b, c) row: sum of the valid frames + (T - len) * tail, / T
__global__ __launch_bounds__(256) void se_pool_kernel(const unsigned short* __restrict__ y, const int* __restrict__ len,
                                                       const float* __restrict__ tail_y, float* __restrict__ pool, int rows,
                                                       int channels, int t, int pitch) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = row / channels, c = row % channels;
  int l = len[b];
  l = l < 0 ? 0 : (l > t ? t : l);
  const unsigned short* src = y + (size_t)row * pitch;
  float s = 0.f;
  for (int g = lane * 8; g < l; g += 64 * 8) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + g);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = g + 2 * j;
      s += (e < l ? bf16_lo(v[j]) : 0.f) + (e + 1 < l ? bf16_hi(v[j]) : 0.f);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) pool[row] = (s + (float)(t - l) * tail_y[c]) / (float)t;
}

// out[b, j] = act( sum_i w[j, i] * in[b, i] ): one wave per output, lanes over the contraction (coalesced weight rows);
// ACT 0: ReLU (first linear), 1: sigmoid (second linear).  Grid: (ceil(n_out / 4), batch).
template <int ACT>
__global__ __launch_bounds__(256) void se_fc_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                     float* __restrict__ out, int n_in, int n_out) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  if (j >= n_out) return;
  const float* x = in + (size_t)b * n_in;
  const float* wr = w + (size_t)j * n_in;
  float s = 0.f;
  for (int i = lane; i < n_in; i += 64) s += wr[i] * x[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) out[(size_t)b * n_out + j] = ACT == 0 ? (s > 0.f ? s : 0.f) : 1.f / (1.f + __expf(-s));
}

// one thread per 8 frames (16 B)
__global__ __launch_bounds__(256) void se_apply_kernel(const unsigned short* __restrict__ y, const unsigned short* __restrict__ r,
                                                        const float* __restrict__ gate, const int* __restrict__ len,
                                                        const float* __restrict__ tail_y, const float* __restrict__ tail_r,
                                                        unsigned short* __restrict__ out, int rows, int channels, int t,
                                                        int pitch_y, int pitch_r, int pitch_out, int relu, int zero_tail, int r_stride) {
  const int groups = pitch_out >> 3;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)rows * groups) return;
  const int row = (int)(idx / groups), g = (int)(idx % groups) * 8;
  const int b = row / channels, c = row % channels;
  int l = len[b];
  l = l < 0 ? 0 : (l > t ? t : l);
  u32x4 o = u32x4{0u, 0u, 0u, 0u};
  if (g < t && (g < l || !zero_tail)) {
    const float gt = gate[row];
    const float ty = tail_y[c], tr = r ? tail_r[c] : 0.f;
    u32x4 yv = u32x4{0u, 0u, 0u, 0u}, rv = u32x4{0u, 0u, 0u, 0u};
    if (g < l) {                                     // rows are zero from their length on, pitches cover the over-read
      if (g + 8 <= pitch_y) yv = *reinterpret_cast<const u32x4*>(y + (size_t)row * pitch_y + g);
      if (r) {
        const unsigned short* rp = r + (size_t)row * pitch_r + (size_t)g * r_stride;
        if (r_stride == 1) {
          if (g + 8 <= pitch_r) rv = *reinterpret_cast<const u32x4*>(rp);
        } else if (r_stride == 2 && 2 * g + 16 <= pitch_r) {
          // the residual branch was computed at the input's frame rate (a 1x1 conv commutes with subsampling): take every other frame
          const u32x4 a0 = *reinterpret_cast<const u32x4*>(rp), a1 = *reinterpret_cast<const u32x4*>(rp + 8);
          rv = u32x4{__builtin_amdgcn_perm(a0[1], a0[0], 0x05040100u), __builtin_amdgcn_perm(a0[3], a0[2], 0x05040100u),
                     __builtin_amdgcn_perm(a1[1], a1[0], 0x05040100u), __builtin_amdgcn_perm(a1[3], a1[2], 0x05040100u)};
        } else {
          unsigned short e8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) e8[j] = (g + j) * r_stride < pitch_r ? rp[(size_t)j * r_stride] : (unsigned short)0;
          rv = u32x4{(unsigned)e8[0] | ((unsigned)e8[1] << 16), (unsigned)e8[2] | ((unsigned)e8[3] << 16),
                     (unsigned)e8[4] | ((unsigned)e8[5] << 16), (unsigned)e8[6] | ((unsigned)e8[7] << 16)};
        }
      }
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf) {
        const int e = g + 2 * j + hlf;
        const float ye = e < l ? (hlf ? bf16_hi(yv[j]) : bf16_lo(yv[j])) : ty;
        const float re = r ? (e < l ? (hlf ? bf16_hi(rv[j]) : bf16_lo(rv[j])) : tr) : 0.f;
        float x = ye * gt + re;
        if (relu) x = x > 0.f ? x : 0.f;
        if (e >= t || (zero_tail && e >= l)) x = 0.f;
        v[2 * j + hlf] = x;
      }
    }
    o = u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
  }
  *reinterpret_cast<u32x4*>(out + (size_t)row * pitch_out + g) = o;
}

}  // namespace ts

extern "C" int ts_se_gate_fwd(const void* y, const int32_t* len, const float* tail_y, int32_t batch, int32_t channels, int32_t t,
                              int32_t pitch, int32_t hidden, const float* w1, const float* w2, float* pool_ws, float* gate,
                              void* stream_) {
  if (!y || !len || !tail_y || !w1 || !w2 || !pool_ws || !gate) return TS_EINVAL;
  if (batch <= 0 || channels <= 0 || hidden <= 0 || t <= 0 || pitch < t || pitch % 8) return TS_EINVAL;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int rows = batch * channels;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::se_pool_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, static_cast<const unsigned short*>(y), len,
                     tail_y, pool_ws, rows, channels, t, pitch);
  // two launches, (outputs / 4) x batch workgroups each: the one-launch form (a workgroup per clip, ts_train_se_gate_fwd) has 32 workgroups pull
  // 1 MB of weights each through one CU's memory port -- 31.7 us against 12.0 us for this pair at 1024 / 128 channels (profiles/round5_c3_se.txt)
  float* const hid = pool_ws + (size_t)batch * channels;           // workspace: [B][C] means, then [B][hidden]
  hipLaunchKernelGGL(ts::se_fc_kernel<0>, dim3((hidden + 3) / 4, batch), dim3(256), 0, stream, pool_ws, w1, hid, channels, hidden);
  hipLaunchKernelGGL(ts::se_fc_kernel<1>, dim3((channels + 3) / 4, batch), dim3(256), 0, stream, hid, w2, gate, hidden, channels);
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_se_apply_fwd(const void* y, const void* r, const float* gate, const int32_t* len, const float* tail_y,
                               const float* tail_r, int32_t batch, int32_t channels, int32_t t, int32_t pitch_y, int32_t pitch_r,
                               int32_t r_stride, int32_t pitch_out, int32_t relu, int32_t zero_tail, void* out, void* stream_) {
  if (!y || !gate || !len || !tail_y || !out || (r && !tail_r)) return TS_EINVAL;
  if (batch <= 0 || channels <= 0 || t <= 0 || pitch_y < t || pitch_out < t || pitch_y % 8 || pitch_out % 8) return TS_EINVAL;
  if (r && (r_stride < 1 || pitch_r < (t - 1) * r_stride + 1 || pitch_r % 8)) return TS_EINVAL;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const long long n = (long long)batch * channels * (pitch_out / 8);
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::se_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                     static_cast<const unsigned short*>(y), static_cast<const unsigned short*>(r), gate, len, tail_y, tail_r,
                     static_cast<unsigned short*>(out), batch * channels, channels, t, pitch_y, pitch_r, pitch_out, relu, zero_tail,
                     r ? r_stride : 1);
  return ts::hip_status(hipGetLastError());
}
