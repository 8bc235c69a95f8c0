// Training-mode encoder kernels (fine-tuning phase 2: encoder unfrozen).  First, unfused version:
// fp32 activations in the reference layout [B][C][T] (contiguous), one launch per reference op, so that every op's
// forward AND backward can be checked against the oracle's autograd.  The pointwise convolutions and their two backward
// products are plain GEMMs and go to rocBLAS (the only library calls in this repo); everything else is hand-written.
//   masked depthwise conv fwd / bwd-data / bwd-weight   quartznet/blocks.py:169-182 (MaskedConv1d, groups = C)
//   masked 1x1 conv fwd / bwd-data / bwd-weight          same class, kernel_size = 1            (rocBLAS sgemm)
//   BatchNorm1d(train) [+ ReLU] fwd / bwd                quartznet/blocks.py:222 (eps 1e-3), statistics over ALL B*T frames (A4)
//   residual add + ReLU fwd / bwd                        quartznet/blocks.py:332-337
#include "ts_common.hpp"

#include "ts_blas.hpp"

namespace ts {

__device__ __forceinline__ int clamp_len(const int* len, int b, int t) {
  if (!len) return t;
  const int l = len[b];
  return l < 0 ? 0 : (l > t ? t : l);
}

constexpr int DW_TILE = 1024;      // output frames per workgroup (forward) / input frames per workgroup (backward-data)
constexpr int DW_KMAX = 128;       // taps cached in LDS

// y[b,c,t] = sum_k w[c,k] * xm[b,c,t*s + k*d - p],  xm = x zeroed from len_in[b] on;  y zeroed from len_out[b] on when given.
// One workgroup = one (clip, channel) row segment of DW_TILE outputs: the input span and the taps are staged in LDS once.
__global__ __launch_bounds__(256) void dw_fwd_kernel(const float* __restrict__ x, const int* __restrict__ len_in,
                                                     const int* __restrict__ len_out, const float* __restrict__ w,
                                                     float* __restrict__ y, int batch, int ch, int t_in, int t_out, int k, int s,
                                                     int d, int p) {
  extern __shared__ float sm[];
  float* const ws = sm;                 // [k]
  float* const xs = sm + DW_KMAX;       // [span]
  const int row = blockIdx.y, b = row / ch, c = row % ch;
  const int t0 = blockIdx.x * DW_TILE;
  const int nt = t_out - t0 < DW_TILE ? t_out - t0 : DW_TILE;
  const int i0 = t0 * s - p;
  const int span = (nt - 1) * s + (k - 1) * d + 1;
  const int li = clamp_len(len_in, b, t_in);
  const float* xr = x + (size_t)row * t_in;
  for (int j = threadIdx.x; j < k; j += 256) ws[j] = w[(size_t)c * k + j];
  for (int e = threadIdx.x; e < span + 14; e += 256) {      // + 14: overreach of the 8-output sliding window (zeros)
    const int i = i0 + e;
    xs[e] = (e < span && i >= 0 && i < li) ? xr[i] : 0.f;
  }
  __syncthreads();
  const int lo = len_out ? clamp_len(len_out, b, t_out) : t_out;
  if (s == 1 && d == 1) {
    // every body layer: 4 consecutive outputs per thread, the input window slides through registers -- per tap one LDS read
    // of the weight (broadcast) and one of the next sample feed 4 FMAs (the one-output form needs 2 reads per FMA)
    for (int t8 = threadIdx.x * 8; t8 < nt; t8 += 2048) {    // 8 outputs per thread: 2 LDS reads per 8 FMAs
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float* xp = xs + t8;                          // xs holds (nt - 1) + k samples (+ zero slack); excess outputs are discarded
      float w[8];
#pragma unroll
      for (int m = 0; m < 7; ++m) w[m] = xp[m];
      for (int j = 0; j < k; ++j) {
        const float wj = ws[j];
        w[7] = xp[j + 7];
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = fmaf(wj, w[m], acc[m]);
#pragma unroll
        for (int m = 0; m < 7; ++m) w[m] = w[m + 1];
      }
#pragma unroll
      for (int m = 0; m < 8; ++m)
        if (t8 + m < nt) y[(size_t)row * t_out + t0 + t8 + m] = t0 + t8 + m < lo ? acc[m] : 0.f;
    }
    return;
  }
  for (int tt = threadIdx.x; tt < nt; tt += 256) {
    float acc = 0.f;
    const float* xp = xs + tt * s;
    for (int j = 0; j < k; ++j) acc = fmaf(ws[j], xp[j * d], acc);
    y[(size_t)row * t_out + t0 + tt] = t0 + tt < lo ? acc : 0.f;
  }
}

// dx[b,c,i] = (i < len_in) ? sum_k w[c,k] * dy[b,c,(i + p - k*d)/s] : 0    (terms with a non-integer or out-of-range index drop out)
// One workgroup = DW_TILE input frames of one row; the dy span that can reach them is staged in LDS.
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(const float* __restrict__ dy, const int* __restrict__ len_in,
                                                          const int* __restrict__ len_out, const float* __restrict__ w,
                                                          float* __restrict__ dx, int batch, int ch, int t_in, int t_out, int k, int s,
                                                          int d, int p) {
  extern __shared__ float sm[];
  float* const ws = sm;
  float* const gs = sm + DW_KMAX;
  const int row = blockIdx.y, b = row / ch, c = row % ch;
  const int i0 = blockIdx.x * DW_TILE;
  const int ni = t_in - i0 < DW_TILE ? t_in - i0 : DW_TILE;
  // n = i + p - j*d ranges over [i0 + p - (k-1)d, i0 + ni - 1 + p]; output frames n / s
  const int n_lo = i0 + p - (k - 1) * d, n_hi = i0 + ni - 1 + p;
  const int g0 = n_lo <= 0 ? 0 : (n_lo + s - 1) / s;
  const int g1 = n_hi < 0 ? -1 : (n_hi / s < t_out - 1 ? n_hi / s : t_out - 1);
  const float* gr = dy + (size_t)row * t_out;
  for (int j = threadIdx.x; j < k; j += 256) ws[j] = w[(size_t)c * k + j];
  const int li = clamp_len(len_in, b, t_in);
  const int lo = len_out ? clamp_len(len_out, b, t_out) : t_out;     // the forward zeroed y from here on: so is its gradient
  if (s == 1 && d == 1) {
    // every body layer: gs2[e] = dy[n_lo + e] with zeros outside the row, dx[i0 + ii] = sum_j w[j] gs2[ii + k - 1 - j];
    // 4 consecutive outputs per thread, the window slides DOWN one sample per tap (2 LDS reads per 4 FMAs, no bounds checks)
    const int n2 = ni + k - 1 + 14;
    for (int e = threadIdx.x; e < n2; e += 256) {
      const int n = n_lo + e - 3;                           // 3 zero samples of slack below the window
      gs[e] = (n >= 0 && n < lo) ? gr[n] : 0.f;
    }
    __syncthreads();
    for (int i8 = threadIdx.x * 8; i8 < ni; i8 += 2048) {   // 8 outputs per thread: 2 LDS reads per 8 FMAs
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float* gp = gs + 3 + i8 + k - 1;                // sample of output i8, tap 0
      float w[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) w[m] = gp[m];
      for (int j = 0; j < k; ++j) {
        const float wj = ws[j];
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = fmaf(wj, w[m], acc[m]);
#pragma unroll
        for (int m = 7; m > 0; --m) w[m] = w[m - 1];
        w[0] = gp[-1 - j];
      }
#pragma unroll
      for (int m = 0; m < 8; ++m)
        if (i8 + m < ni) dx[(size_t)row * t_in + i0 + i8 + m] = i0 + i8 + m < li ? acc[m] : 0.f;
    }
    return;
  }
  for (int e = threadIdx.x; e <= g1 - g0; e += 256) gs[e] = g0 + e < lo ? gr[g0 + e] : 0.f;
  __syncthreads();
  for (int ii = threadIdx.x; ii < ni; ii += 256) {
    const int i = i0 + ii;
    float acc = 0.f;
    if (i < li) {
      if (s == 1) {                                  // every body layer: no divisions in the tap loop
        for (int j = 0; j < k; ++j) {
          const int q = i + p - j * d;
          if (q >= g0 && q <= g1) acc = fmaf(ws[j], gs[q - g0], acc);
        }
      } else {
        for (int j = 0; j < k; ++j) {
          const int n = i + p - j * d;
          if (n >= 0 && n % s == 0) {
            const int q = n / s;
            if (q >= g0 && q <= g1) acc = fmaf(ws[j], gs[q - g0], acc);
          }
        }
      }
    }
    dx[(size_t)row * t_in + i] = acc;
  }
}

// dw[c,j] = sum_{b,t} dy[b,c,t] * xm[b,c,t*s + j*d - p].  One workgroup per (channel, clip group): each clip's dy and x rows
// go through LDS once.  Thread = (group of 4 taps, contiguous slice of the frames); for stride 1 / dilation 1 (every layer but
// the stem and the dilated one) it slides a 4-sample x window through registers: per frame 2 LDS reads feed 4 FMAs (the
// one-tap-per-thread form needs 8).  fp32 partials per clip, fp64 across clips and slices.
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const int* __restrict__ len_in, const int* __restrict__ len_out,
                                                            float* __restrict__ dw, int batch, int ch, int t_in, int t_out, int k, int s,
                                                            int d, int p) {
  extern __shared__ float sm[];
  float* const gs = sm;                       // [t_out]
  float* const xs = sm + t_out + 4;           // [-p .. t_in + 7]: zero margins, so the sliding window needs no bounds checks
  const int xoff = p + 4;                     // xs index of input frame 0
  const int xlen = t_in + 2 * p + 16;
  double* const red = reinterpret_cast<double*>(sm + round_up(t_out + 4 + xlen, 2));   // [256][TG]
  const int c = blockIdx.x;
  constexpr int TG = 8;                       // taps per thread
  const int ng = (k + TG - 1) / TG;           // tap groups
  const int nq = 256 / ng;                    // frame slices
  const int g = threadIdx.x % ng, q = threadIdx.x / ng;
  const bool active = q < nq;
  const int per_q = (t_out + nq - 1) / nq;
  const int t_lo = q * per_q, t_hi = t_lo + per_q < t_out ? t_lo + per_q : t_out;
  double acc[TG];
#pragma unroll
  for (int jj = 0; jj < TG; ++jj) acc[jj] = 0.0;
  const int per = (batch + gridDim.y - 1) / gridDim.y;
  const int b_lo = blockIdx.y * per, b_hi = b_lo + per < batch ? b_lo + per : batch;
  for (int b = b_lo; b < b_hi; ++b) {
    const int li = clamp_len(len_in, b, t_in);
    __syncthreads();
    const int lo = len_out ? clamp_len(len_out, b, t_out) : t_out;
    for (int e = threadIdx.x; e < t_out; e += 256) gs[e] = e < lo ? dy[((size_t)b * ch + c) * t_out + e] : 0.f;
    for (int e = threadIdx.x; e < xlen; e += 256) {
      const int i = e - xoff;
      xs[e] = (i >= 0 && i < li) ? x[((size_t)b * ch + c) * t_in + i] : 0.f;
    }
    __syncthreads();
    if (active) {
      float part[TG];
#pragma unroll
      for (int jj = 0; jj < TG; ++jj) part[jj] = 0.f;
      if (s == 1 && d == 1) {
        // taps TG g .. TG g + TG-1 of frame t read x[t + TG g - p + 0..TG-1]: a window that moves by one sample per frame
        const float* xw = xs + xoff + TG * g - p;
        float w[TG];
#pragma unroll
        for (int jj = 0; jj < TG - 1; ++jj) w[jj] = xw[t_lo + jj];
        for (int t = t_lo; t < t_hi; ++t) {
          const float gv = gs[t];
          w[TG - 1] = xw[t + TG - 1];
#pragma unroll
          for (int jj = 0; jj < TG; ++jj) part[jj] = fmaf(gv, w[jj], part[jj]);
#pragma unroll
          for (int jj = 0; jj < TG - 1; ++jj) w[jj] = w[jj + 1];
        }
      } else {
        for (int t = t_lo; t < t_hi; ++t) {
          const float gv = gs[t];
#pragma unroll
          for (int jj = 0; jj < TG; ++jj) {
            const int i = t * s + (TG * g + jj) * d - p;
            if (TG * g + jj < k && i >= 0 && i < t_in) part[jj] = fmaf(gv, xs[xoff + i], part[jj]);
          }
        }
      }
#pragma unroll
      for (int jj = 0; jj < TG; ++jj) acc[jj] += (double)part[jj];
    }
  }
  __syncthreads();
#pragma unroll
  for (int jj = 0; jj < TG; ++jj) red[threadIdx.x * TG + jj] = active ? acc[jj] : 0.0;
  __syncthreads();
  if (threadIdx.x < k) {
    const int j = threadIdx.x, gj = j / TG, jj = j % TG;
    double tot = 0.0;
    for (int r = 0; r < nq; ++r) tot += red[(r * ng + gj) * TG + jj];
    atomicAdd(dw + (size_t)c * k + j, (float)tot);       // dw is zeroed by the launcher; clips are split over blockIdx.y
  }
}

// y = x with frames >= len[b] zeroed (the re-masking in front of every MaskedConv1d, and of gradients on the way back)
__global__ __launch_bounds__(256) void mask_time_kernel(const float* __restrict__ x, const int* __restrict__ len, float* __restrict__ y,
                                                        int batch, int ch, int t) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)batch * ch * t) return;
  const int tt = (int)(idx % t), b = (int)(idx / ((long long)t * ch));
  y[idx] = tt < clamp_len(len, b, t) ? x[idx] : 0.f;
}

// Per-channel sums over all B*T frames in BN_G clip groups (grid ch x BN_G): part[g][c] = (s1, s2), fp64 accumulation; the
// consumers add the BN_G partials in a fixed order (deterministic, no atomics).
//   MODE 0 (forward statistics):  s1 = sum v,  s2 = sum v^2
//   MODE 1 (backward statistics): g = dy * (y > 0 if relu), xhat = (v - mean) * rstd:  s1 = sum g,  s2 = sum g * xhat
//          -- g and xhat are recomputed here and in the apply kernel instead of being written out and read back twice
constexpr int BN_G = 8;

template <int MODE>
__global__ __launch_bounds__(256) void chan_sums_kernel(const float* __restrict__ a, const float* __restrict__ y, const float* __restrict__ v,
                                                        const float* __restrict__ mean_rstd, double* __restrict__ part, int batch,
                                                        int ch, int t, int relu) {
  __shared__ double r1[256], r2[256];
  const int c = blockIdx.x, grp = blockIdx.y;
  const int per = (batch + BN_G - 1) / BN_G;
  const int b_lo = grp * per, b_hi = b_lo + per < batch ? b_lo + per : batch;
  float mu = 0.f, rs = 0.f;
  if (MODE == 1) { mu = mean_rstd[2 * c]; rs = mean_rstd[2 * c + 1]; }
  double s1 = 0.0, s2 = 0.0;
  for (int b = b_lo; b < b_hi; ++b) {
    const size_t row = ((size_t)b * ch + c) * t;
    for (int i = threadIdx.x; i < t; i += 256) {
      if (MODE == 0) {
        const double va = a[row + i];
        s1 += va;
        s2 += va * va;
      } else {
        const float gv = (relu && !(y[row + i] > 0.f)) ? 0.f : a[row + i];
        const float xh = (v[row + i] - mu) * rs;
        s1 += (double)gv;
        s2 += (double)gv * (double)xh;
      }
    }
  }
  r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[((size_t)grp * ch + c) * 2] = r1[0]; part[((size_t)grp * ch + c) * 2 + 1] = r2[0]; }
}

__device__ __forceinline__ void bn_total(const double* __restrict__ part, int ch, int c, double& s1, double& s2) {
  s1 = 0.0; s2 = 0.0;
#pragma unroll
  for (int g = 0; g < BN_G; ++g) { s1 += part[((size_t)g * ch + c) * 2]; s2 += part[((size_t)g * ch + c) * 2 + 1]; }
}

// BatchNorm(train) forward: stats[c] = (sum v, sum v^2) -> mean, rstd (biased variance, eps), y = gamma*(v-mean)*rstd + beta [ReLU].
// One workgroup per (clip, channel) row segment: the channel's statistics are reduced ONCE per workgroup (not per element) and the
// row is streamed with no index arithmetic.
__global__ __launch_bounds__(256) void bn_fwd_kernel(const float* __restrict__ v, const double* __restrict__ part,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, float* __restrict__ mean_rstd, int batch, int ch, int t,
                                                     float eps, int relu, float* __restrict__ running_mean,
                                                     float* __restrict__ running_var, float momentum,
                                                     long long* __restrict__ num_batches_tracked) {
  __shared__ float sh[2];
  const int row = blockIdx.x, c = row % ch;     // grid: x = (clip, channel) rows, y = 1024-frame chunks
  if (threadIdx.x == 0) {
    const double n = (double)batch * t;
    double s1, s2;
    bn_total(part, ch, c, s1, s2);
    const double mu = s1 / n;
    double var = s2 / n - mu * mu;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    sh[0] = (float)mu; sh[1] = rstd;
    if (row < ch && blockIdx.y == 0) {                           // clip 0's workgroup of this channel publishes the statistics
      mean_rstd[2 * c] = (float)mu; mean_rstd[2 * c + 1] = rstd;
      if (running_mean) {    // nn.BatchNorm1d's update: momentum blend of the batch mean and the UNBIASED batch variance
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * (n / (n > 1.0 ? n - 1.0 : 1.0)));
        if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
      }
    }
  }
  __syncthreads();
  const float mu = sh[0], sc = gamma[c] * sh[1], be = beta[c];
  const size_t base = (size_t)row * t;
  for (int i = blockIdx.y * 1024 + threadIdx.x; i < t && i < (int)(blockIdx.y + 1) * 1024; i += 256) {
    float o = sc * (v[base + i] - mu) + be;
    if (relu) o = o > 0.f ? o : 0.f;
    y[base + i] = o;
  }
}

// dv = gamma*rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy * (y > 0) when relu,  xhat = (v - mean) * rstd; same row-wise shape
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ v, const double* __restrict__ part,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                           float* __restrict__ dv, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           int batch, int ch, int t, int relu) {
  __shared__ float sh[2];
  const int row = blockIdx.x, c = row % ch;     // grid: x = (clip, channel) rows, y = 1024-frame chunks
  if (threadIdx.x == 0) {
    const double n = (double)batch * t;
    double s1, s2;
    bn_total(part, ch, c, s1, s2);
    sh[0] = (float)(s1 / n); sh[1] = (float)(s2 / n);
    if (row < ch && blockIdx.y == 0) { dbeta[c] = (float)s1; dgamma[c] = (float)s2; }
  }
  __syncthreads();
  const float mg = sh[0], mgx = sh[1], mu = mean_rstd[2 * c], rs = mean_rstd[2 * c + 1], k = gamma[c] * rs;
  const size_t base = (size_t)row * t;
  for (int i = blockIdx.y * 1024 + threadIdx.x; i < t && i < (int)(blockIdx.y + 1) * 1024; i += 256) {
    const float g = (relu && !(y[base + i] > 0.f)) ? 0.f : dy[base + i];
    const float xhat = (v[base + i] - mu) * rs;
    dv[base + i] = k * (g - mg - xhat * mgx);
  }
}

// out = relu(a + b); backward: da = db = dout * (out > 0)
__global__ __launch_bounds__(256) void add_relu_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o,
                                                           long long n) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const float s = a[idx] + (b ? b[idx] : 0.f);
  o[idx] = s > 0.f ? s : 0.f;
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out, float* __restrict__ din,
                                                       long long n) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  din[idx] = out[idx] > 0.f ? dout[idx] : 0.f;
}

// sum of `parts` partial [rows] vectors (the per-clip dW of the pointwise backward)
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ parts, float* __restrict__ out, long long rows, int n_parts) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows) return;
  float s = 0.f;
  for (int i = 0; i < n_parts; ++i) s += parts[(size_t)i * rows + idx];
  out[idx] = s;
}

static inline unsigned blocks(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace ts

using namespace ts;
#define TS_STREAM hipStream_t stream = reinterpret_cast<hipStream_t>(stream_); (void)hipGetLastError()

extern "C" int ts_train_dwconv_fwd(const float* x, const int32_t* len_in, const int32_t* len_out, const float* w, float* y, int32_t batch,
                                   int32_t ch, int32_t t_in, int32_t t_out, int32_t k, int32_t stride, int32_t dil, int32_t pad,
                                   void* stream_) {
  if (!x || !w || !y || batch <= 0 || ch <= 0 || t_in <= 0 || t_out <= 0 || k <= 0 || stride <= 0 || dil <= 0) return TS_EINVAL;
  TS_STREAM;
  if (k > DW_KMAX) return TS_EUNSUPPORTED;
  const size_t lds = (DW_KMAX + (size_t)(DW_TILE - 1) * stride + (size_t)(k - 1) * dil + 1 + 16) * sizeof(float);
  if (lds > 64 * 1024) return TS_EUNSUPPORTED;
  hipLaunchKernelGGL(dw_fwd_kernel, dim3((t_out + DW_TILE - 1) / DW_TILE, batch * ch), dim3(256), lds, stream, x, len_in, len_out, w, y,
                     batch, ch, t_in, t_out, k, stride, dil, pad);
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_dwconv_bwd(const float* dy, const float* x, const int32_t* len_in, const int32_t* len_out, const float* w,
                                   float* dx, float* dw, int32_t batch, int32_t ch, int32_t t_in, int32_t t_out, int32_t k,
                                   int32_t stride, int32_t dil, int32_t pad, void* stream_) {
  if (!dy || !x || !w || !dx || !dw || batch <= 0 || ch <= 0 || t_in <= 0 || t_out <= 0 || k <= 0) return TS_EINVAL;
  TS_STREAM;
  if (k > DW_KMAX) return TS_EUNSUPPORTED;
  const size_t lds_d = (DW_KMAX + (size_t)(DW_TILE + (k - 1) * dil) / stride + 2 + 16) * sizeof(float);
  const size_t lds_w = (size_t)round_up(t_out + 4 + t_in + 2 * pad + 16, 2) * sizeof(float) + 256 * 8 * sizeof(double);
  if (k > 256) return TS_EUNSUPPORTED;
  if (lds_d > 64 * 1024 || lds_w > 64 * 1024) return TS_EUNSUPPORTED;
  hipLaunchKernelGGL(dw_bwd_data_kernel, dim3((t_in + DW_TILE - 1) / DW_TILE, batch * ch), dim3(256), lds_d, stream, dy, len_in, len_out, w,
                     dx, batch, ch, t_in, t_out, k, stride, dil, pad);
  if (hipMemsetAsync(dw, 0, sizeof(float) * (size_t)ch * k, stream) != hipSuccess) return TS_EUNSUPPORTED;
  hipLaunchKernelGGL(dw_bwd_weight_kernel, dim3(ch, batch < 8 ? batch : 8), dim3(256), lds_w, stream, dy, x, len_in, len_out, dw, batch, ch,
                     t_in, t_out, k, stride, dil, pad);
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_mask_time(const float* x, const int32_t* len, float* y, int32_t batch, int32_t ch, int32_t t, void* stream_) {
  if (!x || !len || !y || batch <= 0 || ch <= 0 || t <= 0) return TS_EINVAL;
  TS_STREAM;
  hipLaunchKernelGGL(mask_time_kernel, dim3(blocks((long long)batch * ch * t)), dim3(256), 0, stream, x, len, y, batch, ch, t);
  return hip_status(hipGetLastError());
}

// fp32 -> bf16 (round to nearest even): the operand copies of the mixed-precision GEMMs
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long long n) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    *reinterpret_cast<uint2*>(y + i) = uint2{pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)};
  } else {
    for (long long j = i; j < n; ++j) y[j] = (unsigned short)(pack_bf16(x[j], 0.f) & 0xffffu);
  }
}

static rocblas_status gemm_ex(rocblas_handle h, bool bf16, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const void* a,
                              int lda, long long sa, const void* b, int ldb, long long sb, float* c, int ldc, long long sc, int batch) {
  const float one = 1.f, zero = 0.f;
  const rocblas_datatype in = bf16 ? rocblas_datatype_bf16_r : rocblas_datatype_f32_r;
  return rocblas_gemm_strided_batched_ex(h, ta, tb, m, n, k, &one, a, in, lda, sa, b, in, ldb, sb, &zero, c, rocblas_datatype_f32_r, ldc, sc, c,
                                         rocblas_datatype_f32_r, ldc, sc, batch, rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
}

extern "C" int ts_train_cast_bf16(const float* x, void* y, int64_t n, void* stream_) {
  if (!x || !y || n <= 0) return TS_EINVAL;
  if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(y) % 8) return TS_EINVAL;
  TS_STREAM;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks((n + 3) / 4)), dim3(256), 0, stream, x, static_cast<unsigned short*>(y), (long long)n);
  return hip_status(hipGetLastError());
}

// v[b] = W . u[b]   (W [c_out][c_in] row-major, u [B][c_in][t], v [B][c_out][t]); u is expected masked by the caller.
// precision 0: f32 operands; 1: u and w are bf16 (ts_train_cast_bf16), f32 accumulation and result.
extern "C" int ts_train_pwconv_fwd(const void* u, const void* w, float* v, int32_t batch, int32_t c_in, int32_t c_out, int32_t t,
                                   int32_t precision, void* stream_) {
  if (!u || !w || !v || batch <= 0 || c_in <= 0 || c_out <= 0 || t <= 0) return TS_EINVAL;
  if (precision < 0 || precision > 1) return TS_EUNSUPPORTED;
  TS_STREAM;
  rocblas_handle h;
  if (int e = blas(stream, &h)) return e;
  // row-major [c][t] == column-major [t][c]:  V(t x c_out) = U(t x c_in) . Wc(c_in x c_out)
  const rocblas_status st = gemm_ex(h, precision != 0, rocblas_operation_none, rocblas_operation_none, t, c_out, c_in, u, t, (long long)c_in * t,
                                    w, c_in, 0, v, t, (long long)c_out * t, batch);
  return st == rocblas_status_success ? TS_OK : TS_EUNSUPPORTED;
}

// du[b] = W^T . dv[b];  dW = sum_b dv[b] . u[b]^T  (workspace: batch * c_out * c_in floats); precision 1: dv, u, w are bf16
extern "C" int ts_train_pwconv_bwd(const void* dv, const void* u, const void* w, float* du, float* dw, float* workspace, int32_t batch,
                                   int32_t c_in, int32_t c_out, int32_t t, int32_t precision, void* stream_) {
  if (!dv || !u || !w || !du || !dw || !workspace || batch <= 0 || c_in <= 0 || c_out <= 0 || t <= 0) return TS_EINVAL;
  if (precision < 0 || precision > 1) return TS_EUNSUPPORTED;
  TS_STREAM;
  rocblas_handle h;
  if (int e = blas(stream, &h)) return e;
  const bool bf = precision != 0;
  // dU(t x c_in) = dV(t x c_out) . Wc^T(c_out x c_in)
  rocblas_status st = gemm_ex(h, bf, rocblas_operation_none, rocblas_operation_transpose, t, c_in, c_out, dv, t, (long long)c_out * t, w, c_in, 0,
                              du, t, (long long)c_in * t, batch);
  if (st != rocblas_status_success) return TS_EUNSUPPORTED;
  // per clip: dWc_b(c_in x c_out) = U^T(c_in x t) . dV(t x c_out)
  st = gemm_ex(h, bf, rocblas_operation_transpose, rocblas_operation_none, c_in, c_out, t, u, t, (long long)c_in * t, dv, t, (long long)c_out * t,
               workspace, c_in, (long long)c_in * c_out, batch);
  if (st != rocblas_status_success) return TS_EUNSUPPORTED;
  const long long rows = (long long)c_in * c_out;
  hipLaunchKernelGGL(sum_parts_kernel, dim3(blocks(rows)), dim3(256), 0, stream, workspace, dw, rows, batch);
  return hip_status(hipGetLastError());
}

// workspace: 16 * c doubles (8 clip-group partials of 2 sums).  mean_rstd f32 [c][2] is saved for the backward.
extern "C" int ts_train_bn_fwd(const float* v, const float* gamma, const float* beta, float* y, float* mean_rstd, void* workspace,
                               int32_t batch, int32_t ch, int32_t t, float eps, int32_t relu, float* running_mean, float* running_var,
                               float momentum, int64_t* num_batches_tracked, void* stream_) {
  if (!v || !gamma || !beta || !y || !mean_rstd || !workspace || batch <= 0 || ch <= 0 || t <= 0) return TS_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return TS_EINVAL;
  TS_STREAM;
  double* sums = static_cast<double*>(workspace);
  hipLaunchKernelGGL(chan_sums_kernel<0>, dim3(ch, BN_G), dim3(256), 0, stream, v, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, sums, batch, ch, t, 0);
  hipLaunchKernelGGL(bn_fwd_kernel, dim3(batch * ch, (t + 1023) / 1024), dim3(256), 0, stream, v, sums, gamma, beta, y, mean_rstd, batch,
                     ch, t, eps, relu, running_mean, running_var, momentum, reinterpret_cast<long long*>(num_batches_tracked));
  return hip_status(hipGetLastError());
}

// workspace: c * 2 doubles + 2 * batch*ch*t floats (g, xhat)
extern "C" int ts_train_bn_bwd(const float* dy, const float* y, const float* v, const float* gamma, const float* mean_rstd, float* dv,
                               float* dgamma, float* dbeta, void* workspace, int32_t batch, int32_t ch, int32_t t, int32_t relu,
                               void* stream_) {
  if (!dy || !y || !v || !gamma || !mean_rstd || !dv || !dgamma || !dbeta || !workspace || batch <= 0 || ch <= 0 || t <= 0) return TS_EINVAL;
  TS_STREAM;
  const long long n = (long long)batch * ch * t;
  double* sums = static_cast<double*>(workspace);
  hipLaunchKernelGGL(chan_sums_kernel<1>, dim3(ch, BN_G), dim3(256), 0, stream, dy, y, v, mean_rstd, sums, batch, ch, t, relu);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(batch * ch, (t + 1023) / 1024), dim3(256), 0, stream, dy, y, v, sums, gamma, mean_rstd, dv, dgamma, dbeta, batch,
                     ch, t, relu);
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_add_relu_fwd(const float* a, const float* b, float* out, int64_t n, void* stream_) {
  if (!a || !out || n <= 0) return TS_EINVAL;
  TS_STREAM;
  hipLaunchKernelGGL(add_relu_fwd_kernel, dim3(blocks(n)), dim3(256), 0, stream, a, b, out, (long long)n);
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_relu_bwd(const float* dout, const float* out, float* din, int64_t n, void* stream_) {
  if (!dout || !out || !din || n <= 0) return TS_EINVAL;
  TS_STREAM;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks(n)), dim3(256), 0, stream, dout, out, din, (long long)n);
  return hip_status(hipGetLastError());
}
