// The five stages of the mel front end as STANDALONE kernels (reference layout, f32): PreEmphasisFilter, DitherAudio, PowerSpectrum,
// MelScale, FeatureBatchNormalizer called on their own (reference quartznet/transform.py:71-255; the reference's tests call each
// module directly with arbitrary window / FFT sizes, tests/quartznet/test_transform_qn.py:130-260).  FilterbankFeatures never runs
// these: its forward is the fused pair of csrc/frontend.hip (one 512-point FFT kernel + the normaliser).  Generic and simple on
// purpose -- any n_fft (a direct DFT over a twiddle table), any filterbank -- they are not on the hot path.
#include "ts_common.hpp"
#include "ts_philox.hpp"

namespace ts {

// y[b][0] = x[b][0]; y[b][n] = x[b][n] - coeff * x[b][n-1]      (not length-masked: transform.py:136-144)
__global__ void fe_preemph_kernel(const float* __restrict__ x, float* __restrict__ y, int batch, int n, float coeff) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)batch * n) return;
  const int t = (int)(i % n);
  y[i] = t == 0 ? x[i] : x[i] - coeff * x[i - 1];
}

// y = x + dither * N(0, 1) (transform.py:109-118, training mode): the noise of sample k of clip b is the same pure function of
// (seed, b, k) the fused front end draws inside its sample load (csrc/frontend.hip dither_noise; oracle/augment.py restates it)
__global__ void fe_dither_kernel(const float* __restrict__ x, float* __restrict__ y, int batch, int n, float dither, unsigned long long seed) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // pair of samples: one Philox draw, one Box-Muller pair
  const int pairs = (n + 1) / 2;
  if (q >= (long long)batch * pairs) return;
  const int b = (int)(q / pairs), g = (int)(q % pairs);
  const Philox4 r = philox(seed, PHILOX_DITHER, ((unsigned long long)(unsigned)b << 32) | (unsigned)g);
  float n0, n1;
  normal2(r.v[0], r.v[1], n0, n1);
  const size_t i = (size_t)b * n + 2 * g;
  y[i] = x[i] + dither * n0;
  if (2 * g + 1 < n) y[i + 1] = x[i + 1] + dither * n1;
}

// |STFT|^2 as torch.stft(center=True, pad_mode="reflect") computes it: frame f covers the reflect-padded samples [f hop, f hop + n_fft),
// window = `window` (n_fft entries: the win_length window centred in zeros).  One workgroup per (clip, frame): the windowed frame is
// staged in LDS, thread k sums the DFT bin k over a cos/sin table of n_fft entries (index k j mod n_fft: exact phase reduction).
// COMPLEX: out[b][k][f][2] = (re, im) of the transform itself (what the reference's convolution_stft returns, blocks.py:38-91)
template <bool COMPLEX>
__global__ void fe_power_kernel(const float* __restrict__ x, const float* __restrict__ window, const float* __restrict__ twiddle,
                                float* __restrict__ out, int n, int n_fft, int hop, int frames) {
  extern __shared__ float frame[];                 // [n_fft]
  const int b = blockIdx.y, f = blockIdx.x;
  const int half = n_fft / 2;
  for (int j = threadIdx.x; j < n_fft; j += blockDim.x) {
    int s = f * hop + j - half;                    // index into the unpadded waveform
    if (s < 0) s = -s;                             // reflect (no edge repeat), as F.pad(mode="reflect")
    if (s >= n) s = 2 * (n - 1) - s;
    s = s < 0 ? 0 : (s >= n ? n - 1 : s);
    frame[j] = x[(size_t)b * n + s] * window[j];
  }
  __syncthreads();
  const int n_freq = half + 1;
  for (int k = threadIdx.x; k < n_freq; k += blockDim.x) {
    float re = 0.f, im = 0.f;
    int ph = 0;                                    // k * j mod n_fft
    for (int j = 0; j < n_fft; ++j) {
      const float v = frame[j];
      re += v * twiddle[2 * ph];
      im -= v * twiddle[2 * ph + 1];
      ph += k;
      if (ph >= n_fft) ph -= n_fft;
    }
    const size_t o = ((size_t)b * n_freq + k) * frames + f;
    if constexpr (COMPLEX) { out[2 * o] = re; out[2 * o + 1] = im; }
    else out[o] = re * re + im * im;
  }
}

// out[b][m][t] = log(sum_f fb[m][f] * x[b][f][t] + 2^-24)  (log_scale) or the plain product      (transform.py:243-255)
__global__ void fe_mel_kernel(const float* __restrict__ x, const float* __restrict__ fb, float* __restrict__ out, int batch, int n_freq,
                              int n_mels, int t, int log_scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)batch * n_mels * t) return;
  const int tt = (int)(i % t);
  const int m = (int)((i / t) % n_mels);
  const int b = (int)(i / ((long long)t * n_mels));
  const float* xr = x + (size_t)b * n_freq * t + tt;
  const float* w = fb + (size_t)m * n_freq;
  float acc = 0.f;
  for (int f = 0; f < n_freq; ++f) acc += w[f] * xr[(size_t)f * t];
  out[i] = log_scale ? logf(acc + 5.9604644775390625e-08f) : acc;
}

// normalize_tensor with a mask (blocks.py:136-149, quirk A1): per (clip, feature) row, over the first len[b] frames:
// mean = sum_valid / N; std = sqrt((sum_valid (x - mean)^2 + (T - N) mean^2) / N); out = (x - mean) / (std + guard), 0 beyond the length.
// One wave per row, f64 accumulation.
__global__ void fe_normalize_kernel(const float* __restrict__ x, const int* __restrict__ len, float* __restrict__ out, int rows, int features,
                                    int t, float guard) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  int n = len[row / features];
  n = n < 0 ? 0 : (n > t ? t : n);
  const float* xr = x + (size_t)row * t;
  double s = 0.0;
  for (int i = lane; i < n; i += 64) s += xr[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const double mean = s / (double)n;               // n == 0: NaN, as in the reference (0 / 0); those rows are fully masked below
  double q = 0.0;
  for (int i = lane; i < n; i += 64) { const double d = (double)xr[i] - mean; q += d * d; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  q += (double)(t - n) * mean * mean;              // the zero-filled padded frames are part of the reference's numerator
  const float fmean = (float)mean;
  const float inv = 1.f / ((float)sqrt(q / (double)n) + guard);
  float* o = out + (size_t)row * t;
  for (int i = lane; i < t; i += 64) o[i] = i < n ? (xr[i] - fmean) * inv : 0.f;
}

}  // namespace ts

extern "C" int ts_fe_preemph(const float* x, float* y, int32_t batch, int32_t n, float coeff, void* stream) {
  if (!x || !y || batch <= 0 || n <= 0) return TS_EINVAL;
  const long long total = (long long)batch * n;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::fe_preemph_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, batch, n, coeff);
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_fe_dither(const float* x, float* y, int32_t batch, int32_t n, float dither, uint64_t seed, void* stream) {
  if (!x || !y || batch <= 0 || n <= 0) return TS_EINVAL;
  const long long total = (long long)batch * ((n + 1) / 2);
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::fe_dither_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, batch, n, dither,
                     (unsigned long long)seed);
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_fe_power_spectrum(const float* x, const float* window, const float* twiddle, float* out, int32_t batch, int32_t n,
                                    int32_t n_fft, int32_t hop, void* stream) {
  if (!x || !window || !twiddle || !out || batch <= 0 || n <= 0 || n_fft < 2 || n_fft > 8192 || hop <= 0) return TS_EINVAL;
  if (n_fft / 2 >= n) return TS_EINVAL;            // reflect padding needs n_fft / 2 < n (torch.stft raises as well)
  const int frames = 1 + (n + 2 * (n_fft / 2) - n_fft) / hop;      // torch.stft(center=True): n / hop + 1 for even n_fft, (n - 1) / hop + 1 for odd
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::fe_power_kernel<false>, dim3(frames, batch), dim3(256), (size_t)n_fft * sizeof(float), (hipStream_t)stream, x, window, twiddle,
                     out, n, n_fft, hop, frames);
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_fe_stft(const float* x, const float* window, const float* twiddle, float* out, int32_t batch, int32_t n, int32_t n_fft,
                          int32_t hop, void* stream) {
  if (!x || !window || !twiddle || !out || batch <= 0 || n <= 0 || n_fft < 2 || n_fft > 8192 || hop <= 0) return TS_EINVAL;
  if (n_fft / 2 >= n) return TS_EINVAL;
  const int frames = 1 + (n + 2 * (n_fft / 2) - n_fft) / hop;      // torch.stft(center=True): n / hop + 1 for even n_fft, (n - 1) / hop + 1 for odd
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::fe_power_kernel<true>, dim3(frames, batch), dim3(256), (size_t)n_fft * sizeof(float), (hipStream_t)stream, x, window, twiddle,
                     out, n, n_fft, hop, frames);
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_fe_mel(const float* x, const float* fb, float* out, int32_t batch, int32_t n_freq, int32_t n_mels, int32_t t,
                         int32_t log_scale, void* stream) {
  if (!x || !fb || !out || batch <= 0 || n_freq <= 0 || n_mels <= 0 || t <= 0) return TS_EINVAL;
  const long long total = (long long)batch * n_mels * t;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::fe_mel_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, fb, out, batch, n_freq,
                     n_mels, t, log_scale);
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_fe_normalize(const float* x, const int32_t* len, float* out, int32_t batch, int32_t features, int32_t t, float guard,
                               void* stream) {
  if (!x || !len || !out || batch <= 0 || features <= 0 || t <= 0) return TS_EINVAL;
  const int rows = batch * features;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::fe_normalize_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, len, out, rows, features, t,
                     guard);
  return ts::hip_status(hipGetLastError());
}
