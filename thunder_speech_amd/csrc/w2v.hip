// wav2vec2 waveform normalisation (reference huggingface/transform.py:34-55 -> blocks.py:118-153 normalize_tensor):
//   mask_input = 0: (x - mean) / sqrt(var_unbiased + guard)                      over all T samples
//   mask_input = 1: x' = x zeroed beyond len;  mean = sum(x') / N;  sigma = sqrt( sum_over_ALL_T (x' - mean)^2 / N )
//                   (the padded samples each add mean^2, as in quirk A1);  (x' - mean) / (sigma + guard), zero beyond len
// Two launches: per-(clip, chunk) partial sums in fp64, then every workgroup reduces its clip's partials and writes its chunk.
#include "ts_common.hpp"

namespace ts {

constexpr int W2V_CHUNKS = 64;

__global__ __launch_bounds__(256) void w2v_partial_kernel(const float* __restrict__ x, const int* __restrict__ len,
                                                          double* __restrict__ partial, int t, int mask_input) {
  __shared__ double s1[256], s2[256];
  const int b = blockIdx.y, c = blockIdx.x;
  const int n = mask_input ? (len[b] < t ? (len[b] < 0 ? 0 : len[b]) : t) : t;
  const long long per = ((long long)t + W2V_CHUNKS - 1) / W2V_CHUNKS;
  const long long lo = c * per, hi = lo + per < n ? lo + per : n;
  const float* row = x + (size_t)b * t;
  double a1 = 0.0, a2 = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const double v = row[i];
    a1 += v;
    a2 += v * v;
  }
  s1[threadIdx.x] = a1; s2[threadIdx.x] = a2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[((size_t)b * W2V_CHUNKS + c) * 2] = s1[0];
    partial[((size_t)b * W2V_CHUNKS + c) * 2 + 1] = s2[0];
  }
}

__global__ __launch_bounds__(256) void w2v_apply_kernel(const float* __restrict__ x, const int* __restrict__ len,
                                                        const double* __restrict__ partial, float* __restrict__ y, int t,
                                                        int mask_input, float guard) {
  __shared__ float mu_s, rs_s;
  const int b = blockIdx.y, c = blockIdx.x;
  const int n = mask_input ? (len[b] < t ? (len[b] < 0 ? 0 : len[b]) : t) : t;
  if (threadIdx.x == 0) {
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < W2V_CHUNKS; ++i) { s1 += partial[((size_t)b * W2V_CHUNKS + i) * 2]; s2 += partial[((size_t)b * W2V_CHUNKS + i) * 2 + 1]; }
    const double mu = n > 0 ? s1 / n : 0.0;
    double centered = s2 - (double)n * mu * mu;                      // sum over the valid samples of (x - mu)^2
    centered = centered < 0.0 ? 0.0 : centered;
    double rs;
    if (mask_input) {
      const double sigma = sqrt((centered + (double)(t - n) * mu * mu) / (n > 0 ? n : 1));
      rs = 1.0 / (sigma + (double)guard);
    } else {
      rs = 1.0 / sqrt(centered / (t > 1 ? t - 1 : 1) + (double)guard);
    }
    mu_s = (float)mu; rs_s = (float)rs;
  }
  __syncthreads();
  const float mu = mu_s, rs = rs_s;
  const long long per = ((long long)t + W2V_CHUNKS - 1) / W2V_CHUNKS;
  const long long lo = c * per, hi = lo + per < t ? lo + per : t;
  for (long long i = lo + threadIdx.x; i < hi; i += 256)
    y[(size_t)b * t + i] = i < n ? (x[(size_t)b * t + i] - mu) * rs : 0.f;
}

}  // namespace ts

extern "C" int64_t ts_w2v_workspace_bytes(int32_t batch) { return batch > 0 ? (int64_t)batch * ts::W2V_CHUNKS * 2 * sizeof(double) : TS_EINVAL; }

extern "C" int ts_w2v_preprocess(const float* wave, const int32_t* wave_len, int32_t batch, int32_t n_samples, int32_t mask_input,
                                 float div_guard, float* out, void* workspace, void* stream_) {
  if (!wave || !out || !workspace || batch <= 0 || n_samples <= 0 || (mask_input && !wave_len)) return TS_EINVAL;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::w2v_partial_kernel, dim3(ts::W2V_CHUNKS, batch), dim3(256), 0, stream, wave, wave_len,
                     static_cast<double*>(workspace), n_samples, mask_input);
  hipLaunchKernelGGL(ts::w2v_apply_kernel, dim3(ts::W2V_CHUNKS, batch), dim3(256), 0, stream, wave, wave_len,
                     static_cast<const double*>(workspace), out, n_samples, mask_input, div_guard);
  return ts::hip_status(hipGetLastError());
}
