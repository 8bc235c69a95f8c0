// Training-mode encoder kernels (fine-tuning phase 2: encoder unfrozen): one launch per reference op and
// direction, so that every op's forward AND backward can be checked against the oracle's autograd.  Activations are [B][C][pitch]
// rows (time contiguous, pitch a multiple of 8 elements, 16-byte aligned rows; columns >= T are scratch) in one of two element
// types, selected per call by `act`: 0 = f32 (the reference's arithmetic), 1 = bf16 storage with f32 arithmetic inside every
// kernel (mixed precision: half the activation traffic, bf16 MFMA GEMMs; parameters, gradients of parameters, statistics stay
// f32).  The pointwise convolutions and their two backward products are plain GEMMs and go to rocBLAS; everything else is
// hand-written.
//   masked depthwise conv fwd / bwd-data / bwd-weight   quartznet/blocks.py:169-182 (MaskedConv1d, groups = C)
//   masked 1x1 conv fwd / bwd-data / bwd-weight          same class, kernel_size = 1            (rocBLAS sgemm)
//   BatchNorm1d(train) [+ ReLU] fwd / bwd                quartznet/blocks.py:222 (eps 1e-3), statistics over ALL B*T frames (A4)
//   residual add + ReLU fwd / bwd                        quartznet/blocks.py:332-337
#include "ts_common.hpp"

#include <cstdlib>


namespace ts {

__device__ __forceinline__ int clamp_len(const int* len, int b, int t) {
  if (!len) return t;
  const int l = len[b];
  return l < 0 ? 0 : (l > t ? t : l);
}

// element access of the two activation types (f32 / bf16 bits)
typedef unsigned short bf16_t;
__device__ __forceinline__ float ldf(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ float ldf(const bf16_t* p, size_t i) { return bf16_to_f32(p[i]); }
__device__ __forceinline__ void stf(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void stf(bf16_t* p, size_t i, float v) { p[i] = (bf16_t)(pack_bf16(v, 0.f) & 0xffffu); }
// 8 consecutive elements of a 16-byte aligned row position
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  const u32x4 a = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[2 * j] = bf16_lo(a[j]); v[2 * j + 1] = bf16_hi(a[j]); }
}
// Activation rows leave with streaming (nontemporal) stores: a training launch writes 8-16 MB that the NEXT launch reads, possibly on another
// XCD, so the lines have to reach memory anyway -- streaming them out while the kernel runs beats leaving them dirty in the XCD's L2 for the
// end-of-kernel write-back the next launch waits for (measured on the 1x1 products: -10 % per launch, profiles/round6_pw_tile.txt).
#ifndef TS_TRAIN_NT
#define TS_TRAIN_NT 0
#endif
__device__ __forceinline__ void st16(u32x4* p, u32x4 v) {
#if TS_TRAIN_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ void st16(f32x4* p, f32x4 v) {
#if TS_TRAIN_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  st16(reinterpret_cast<f32x4*>(p), f32x4{v[0], v[1], v[2], v[3]});
  st16(reinterpret_cast<f32x4*>(p + 4), f32x4{v[4], v[5], v[6], v[7]});
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  st16(reinterpret_cast<u32x4*>(p), u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])});
}

constexpr int DW_TILE = 1024;      // output frames per workgroup (forward) / input frames per workgroup (backward-data)
constexpr int DW_KMAX = 128;       // taps cached in LDS

// 8 consecutive outputs of a stride-1 FIR out of LDS: acc[m] = sum_j taps[j] * xp[m + j], j < k8 (a multiple of 8; taps are
// zero-padded).  Eight taps at a time: two 16-byte reads fetch the taps (same address in every lane: broadcast), two more the
// next 8 samples of the window; the 64 FMAs of the step index a 16-register window statically -- no per-tap register shifts,
// 4 LDS reads per 64 FMAs (the one-tap-at-a-time loop needs 16 reads and 56 moves).  xp and taps must be 16-byte aligned.
__device__ __forceinline__ void fir8(const float* __restrict__ xp, const float* __restrict__ taps, int k8, float (&acc)[8]) {
  float win[16];
  {
    const f32x4 a = *reinterpret_cast<const f32x4*>(xp), b = *reinterpret_cast<const f32x4*>(xp + 4);
    win[0] = a[0]; win[1] = a[1]; win[2] = a[2]; win[3] = a[3]; win[4] = b[0]; win[5] = b[1]; win[6] = b[2]; win[7] = b[3];
  }
  for (int j0 = 0; j0 < k8; j0 += 8) {
    const f32x4 c = *reinterpret_cast<const f32x4*>(xp + j0 + 8), d = *reinterpret_cast<const f32x4*>(xp + j0 + 12);
    win[8] = c[0]; win[9] = c[1]; win[10] = c[2]; win[11] = c[3]; win[12] = d[0]; win[13] = d[1]; win[14] = d[2]; win[15] = d[3];
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(taps + j0), w1 = *reinterpret_cast<const f32x4*>(taps + j0 + 4);
    const float w[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[m] = fmaf(w[jj], win[m + jj], acc[m]);
#pragma unroll
    for (int m = 0; m < 8; ++m) win[m] = win[m + 8];
  }
}

// y[b,c,t] = sum_k w[c,k] * xm[b,c,t*s + k*d - p],  xm = x zeroed from len_in[b] on;  y zeroed from len_out[b] on when given.
// One workgroup = one (clip, channel) row segment of DW_TILE outputs: the input span and the taps are staged in LDS once.
template <class T>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const T* __restrict__ x, const int* __restrict__ len_in,
                                                     const int* __restrict__ len_out, const float* __restrict__ w,
                                                     T* __restrict__ y, int batch, int ch, int t_in, int t_out, int k, int s,
                                                     int d, int p, int pitch_in, int pitch_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* const ws = sm;                 // [k], zero-padded to a multiple of 8
  float* const xs = sm + DW_KMAX;       // [span]
  const int row = blockIdx.y, b = row / ch, c = row % ch;
  const int t0 = blockIdx.x * DW_TILE;
  const int nt = t_out - t0 < DW_TILE ? t_out - t0 : DW_TILE;
  const int i0 = t0 * s - p;
  const int span = (nt - 1) * s + (k - 1) * d + 1;
  const int li = clamp_len(len_in, b, t_in);
  const T* xr = x + (size_t)row * pitch_in;
  T* const yr = y + (size_t)row * pitch_out;
  const int k8 = (k + 7) & ~7;
  for (int j = threadIdx.x; j < k8; j += 256) ws[j] = j < k ? w[(size_t)c * k + j] : 0.f;
  for (int e = threadIdx.x; e < span + 24; e += 256) {      // + 24: overreach of the 8-output x 8-tap window (zeros)
    const int i = i0 + e;
    xs[e] = (e < span && i >= 0 && i < li) ? ldf(xr, i) : 0.f;
  }
  __syncthreads();
  const int lo = len_out ? clamp_len(len_out, b, t_out) : t_out;
  if (s == 1 && d == 1) {
    // every body layer: 8 consecutive outputs per thread through the 8-tap micro-kernel
    for (int t8 = threadIdx.x * 8; t8 < nt; t8 += 2048) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      fir8(xs + t8, ws, k8, acc);                          // xs holds (nt - 1) + k samples (+ zero slack); excess outputs are discarded
#pragma unroll
      for (int m = 0; m < 8; ++m)
        if (t8 + m < nt) stf(yr, t0 + t8 + m, t0 + t8 + m < lo ? acc[m] : 0.f);
    }
    return;
  }
  for (int tt = threadIdx.x; tt < nt; tt += 256) {
    float acc = 0.f;
    const float* xp = xs + tt * s;
    for (int j = 0; j < k; ++j) acc = fmaf(ws[j], xp[j * d], acc);
    stf(yr, t0 + tt, t0 + tt < lo ? acc : 0.f);
  }
}

// dx[b,c,i] = (i < len_in) ? sum_k w[c,k] * dy[b,c,(i + p - k*d)/s] : 0    (terms with a non-integer or out-of-range index drop out)
// One workgroup = DW_TILE input frames of one row; the dy span that can reach them is staged in LDS.
template <class T>
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(const T* __restrict__ dy, const int* __restrict__ len_in,
                                                          const int* __restrict__ len_out, const float* __restrict__ w,
                                                          T* __restrict__ dx, int batch, int ch, int t_in, int t_out, int k, int s,
                                                          int d, int p, int pitch_in, int pitch_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* const ws = sm;
  float* const gs = sm + DW_KMAX;
  const int row = blockIdx.y, b = row / ch, c = row % ch;
  const int i0 = blockIdx.x * DW_TILE;
  const int ni = t_in - i0 < DW_TILE ? t_in - i0 : DW_TILE;
  // n = i + p - j*d ranges over [i0 + p - (k-1)d, i0 + ni - 1 + p]; output frames n / s
  const int n_lo = i0 + p - (k - 1) * d, n_hi = i0 + ni - 1 + p;
  const int g0 = n_lo <= 0 ? 0 : (n_lo + s - 1) / s;
  const int g1 = n_hi < 0 ? -1 : (n_hi / s < t_out - 1 ? n_hi / s : t_out - 1);
  const T* gr = dy + (size_t)row * pitch_out;
  T* const dxr = dx + (size_t)row * pitch_in;
  for (int j = threadIdx.x; j < k; j += 256) ws[j] = w[(size_t)c * k + j];
  const int li = clamp_len(len_in, b, t_in);
  const int lo = len_out ? clamp_len(len_out, b, t_out) : t_out;     // the forward zeroed y from here on: so is its gradient
  if (s == 1 && d == 1) {
    // every body layer: the same FIR with the taps FLIPPED: dx[i0 + ii] = sum_j' w[k-1-j'] gs[8 + ii + j'], gs[8 + q] = dy[n_lo + q]
    // (zeros outside the row; 8 zero samples in front keep the window reads aligned, the flipped taps are padded at the END)
    const int k8 = (k + 7) & ~7;
    __syncthreads();                                        // ws is rewritten below (the loop above filled it in forward order)
    for (int j = threadIdx.x; j < k8; j += 256) ws[j] = j < k ? w[(size_t)c * k + (k - 1 - j)] : 0.f;
    const int n2 = ni + k - 1 + 8 + 24;
    for (int e = threadIdx.x; e < n2; e += 256) {
      const int n = n_lo + e - 8;
      gs[e] = (n >= 0 && n < lo) ? ldf(gr, n) : 0.f;
    }
    __syncthreads();
    for (int i8 = threadIdx.x * 8; i8 < ni; i8 += 2048) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      fir8(gs + 8 + i8, ws, k8, acc);
#pragma unroll
      for (int m = 0; m < 8; ++m)
        if (i8 + m < ni) stf(dxr, i0 + i8 + m, i0 + i8 + m < li ? acc[m] : 0.f);
    }
    return;
  }
  for (int e = threadIdx.x; e <= g1 - g0; e += 256) gs[e] = g0 + e < lo ? ldf(gr, g0 + e) : 0.f;
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
    stf(dxr, i, acc);
  }
}

// dw[c,j] = sum_{b,t} dy[b,c,t] * xm[b,c,t*s + j*d - p].  One workgroup per (channel, clip group): each clip's dy and x rows
// go through LDS once.  Thread = (group of 4 taps, contiguous slice of the frames); for stride 1 / dilation 1 (every layer but
// the stem and the dilated one) it slides a 4-sample x window through registers: per frame 2 LDS reads feed 4 FMAs (the
// one-tap-per-thread form needs 8).  fp32 partials per clip, fp64 across clips and slices.
template <class T>
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const int* __restrict__ len_in, const int* __restrict__ len_out,
                                                            float* __restrict__ dw, int batch, int ch, int t_in, int t_out, int k, int s,
                                                            int d, int p, int pitch_in, int pitch_out, float* __restrict__ det_part) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* const gs = sm;                       // [t_out], zero-padded to the 8-frame steps of the slices
  const int gpad = round_up(t_out + 16, 4);
  float* const xs = sm + gpad;                // [-p .. t_in + ...]: zero margins, so the sliding window needs no bounds checks
  const int xoff = p + 4;                     // xs index of input frame 0
  const int xlen = round_up(t_in + 2 * p + 48, 2);
  double* const red = reinterpret_cast<double*>(sm + gpad + xlen);   // [256][TG]
  const int c = blockIdx.x;
  constexpr int TG = 8;                       // taps per thread
  const int ng = (k + TG - 1) / TG;           // tap groups
  const int nq = 256 / ng;                    // frame slices
  const int g = threadIdx.x % ng, q = threadIdx.x / ng;
  const bool active = q < nq;
  const int per_q = (s == 1 && d == 1) ? round_up((t_out + nq - 1) / nq, 8) : (t_out + nq - 1) / nq;   // 8-frame steps (vector LDS reads)
  const int t_lo = q * per_q < t_out ? q * per_q : t_out, t_hi = t_lo + per_q < t_out ? t_lo + per_q : t_out;
  double acc[TG];
#pragma unroll
  for (int jj = 0; jj < TG; ++jj) acc[jj] = 0.0;
  const int per = (batch + gridDim.y - 1) / gridDim.y;
  const int b_lo = blockIdx.y * per, b_hi = b_lo + per < batch ? b_lo + per : batch;
  for (int b = b_lo; b < b_hi; ++b) {
    const int li = clamp_len(len_in, b, t_in);
    __syncthreads();
    const int lo = len_out ? clamp_len(len_out, b, t_out) : t_out;
    for (int e = threadIdx.x; e < gpad; e += 256) gs[e] = e < lo ? ldf(dy, ((size_t)b * ch + c) * pitch_out + e) : 0.f;
    for (int e = threadIdx.x; e < xlen; e += 256) {
      const int i = e - xoff;
      xs[e] = (i >= 0 && i < li) ? ldf(x, ((size_t)b * ch + c) * pitch_in + i) : 0.f;
    }
    __syncthreads();
    if (active) {
      float part[TG];
#pragma unroll
      for (int jj = 0; jj < TG; ++jj) part[jj] = 0.f;
      if (s == 1 && d == 1) {
        // taps TG g .. TG g + TG-1 of frame t read x[t + TG g - p + 0..TG-1].  8 frames per step: two 16-byte reads fetch the
        // gradients, two more the next 8 samples of a 16-register window; the 64 FMAs index the window statically (4 LDS reads
        // per 64 FMAs instead of 16).  Frames >= t_out hold zero gradients, so whole 8-frame steps are safe.
        const float* xw = xs + xoff + TG * g - p;
        float win[16];
        if (t_lo < t_hi) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(xw + t_lo), b2 = *reinterpret_cast<const f32x4*>(xw + t_lo + 4);
          win[0] = a[0]; win[1] = a[1]; win[2] = a[2]; win[3] = a[3]; win[4] = b2[0]; win[5] = b2[1]; win[6] = b2[2]; win[7] = b2[3];
        }
        for (int t = t_lo; t < t_hi; t += 8) {
          const f32x4 c4 = *reinterpret_cast<const f32x4*>(xw + t + 8), d4 = *reinterpret_cast<const f32x4*>(xw + t + 12);
          win[8] = c4[0]; win[9] = c4[1]; win[10] = c4[2]; win[11] = c4[3]; win[12] = d4[0]; win[13] = d4[1]; win[14] = d4[2]; win[15] = d4[3];
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(gs + t), g1 = *reinterpret_cast<const f32x4*>(gs + t + 4);
          const float gv[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
#pragma unroll
          for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int jj = 0; jj < TG; ++jj) part[jj] = fmaf(gv[m], win[m + jj], part[jj]);
#pragma unroll
          for (int m = 0; m < 8; ++m) win[m] = win[m + 8];
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
    // dw accumulates (zero on entry for a plain gradient); clips are split over blockIdx.y (deterministic mode: ordered partials, see det_reduce_kernel)
    if (det_part) det_part[((size_t)blockIdx.y * ch + c) * (k + 2) + j] = (float)tot;
    else atomicAdd(dw + (size_t)c * k + j, (float)tot);
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// "Same" depthwise convolutions (stride 1, dilation 1, odd k, pad = (k-1)/2, even channel count: every body layer of QuartzNet /
// Citrinet) -- the VALU-bound part of the training step.  One WAVEFRONT owns TWO adjacent channel rows of one clip and every
// product is a packed-f32 FMA over the pair (v_pk_fma_f32: lane-half 0 = channel c, lane-half 1 = channel c + 1), which doubles
// the FMA rate of the one-row form; each lane computes 8 consecutive frames from a 16-entry register window (fir_pair), the taps
// are broadcast reads of a per-wave LDS copy, so a step of 64 packed FMAs costs 8 LDS reads.  No workgroup barrier in the
// streaming part: a wave stages its own rows (LDS operations of one wave execute in order).
// LDS layout of a staged row pair: float2 samples, 2-sample chunks dealt round-robin over 4 sub-arrays, so that the four 16-byte
// reads of a window step are lane-contiguous in each sub-array (a lane's window starts 8 samples after its neighbour's).
// ----------------------------------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int PT = 512;                                        // frames per wave tile
struct __attribute__((aligned(16))) v2fx2 { v2f a, b; };

// Staging of frames [a0, a0 + n) of rows ra / rb (a0 and n multiples of 8, n <= 1024; zeros outside [0, lim)) in two halves, so that
// the global loads of the NEXT row pair are in flight while the current one is filtered:
//   stage_fetch  a lane loads 8 aligned frames of both rows, twice (frames a0 + 8 lane and a0 + 8 (lane + 64)), raw, into registers
//   stage_write  converts them, zeroes what lies outside [0, lim) and writes four 16-byte chunks per 8 frames, one per sub-array
// The FIR origin is therefore 8-aligned; the callers shift their TAPS by (t0 - p) & 7 instead of the samples.
template <class T> struct Raw8;
template <> struct Raw8<float> { f32x4 lo, hi; };
template <> struct Raw8<bf16_t> { u32x4 v; };
__device__ __forceinline__ void raw_load(Raw8<float>& r, const float* p) { r.lo = *reinterpret_cast<const f32x4*>(p); r.hi = *reinterpret_cast<const f32x4*>(p + 4); }
__device__ __forceinline__ void raw_load(Raw8<bf16_t>& r, const bf16_t* p) { r.v = *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void raw_get(const Raw8<float>& r, float (&v)[8]) {
  v[0] = r.lo[0]; v[1] = r.lo[1]; v[2] = r.lo[2]; v[3] = r.lo[3]; v[4] = r.hi[0]; v[5] = r.hi[1]; v[6] = r.hi[2]; v[7] = r.hi[3];
}
__device__ __forceinline__ void raw_get(const Raw8<bf16_t>& r, float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[2 * j] = bf16_lo(r.v[j]); v[2 * j + 1] = bf16_hi(r.v[j]); }
}
template <class T> struct Staged { Raw8<T> a[2], b[2]; };

// PH (phase split, dilation-2 layers): the "pair" is ONE row seen as (even frame, odd frame) entries -- y[2s + ph] = sum_u w[u] x[2 (s + u)
// + ph - pad] is a dilation-1 convolution of each phase with the same taps -- so entry e = frames (2e, 2e + 1): 8 entries = 16 consecutive
// frames of the row (two raw loads), ra = the row, rb unused; a0, n, the FIR and the tiles count entries, `lim` stays in frames.
template <class T, bool PH = false>
__device__ __forceinline__ void stage_fetch(Staged<T>& st, const T* ra, const T* rb, int a0, int n, int lim, int lane) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int u = lane + 64 * it;
    if (PH) {
      const int i16 = 2 * a0 + 16 * u;
      if (8 * u < n && i16 >= 0 && i16 < lim) { raw_load(st.a[it], ra + i16); if (i16 + 8 < lim) raw_load(st.b[it], ra + i16 + 8); }
    } else {
      const int i8 = a0 + 8 * u;
      if (8 * u < n && i8 >= 0 && i8 < lim) { raw_load(st.a[it], ra + i8); raw_load(st.b[it], rb + i8); }
    }
  }
}
// Input transform of the rows being staged: the BatchNorm(train) [+ ReLU] of the PREVIOUS repeat, y = relu?(v * scale + shift), applied on
// the fly so that y is never stored (train_ops.SubBlock hands the un-normalised v of a repeat to the next one).  on == false: identity.
struct RowAffine { float sa, ha, sb, hb; bool relu, on; };
constexpr int BN_G = 8;                                   // clip groups of the BatchNorm partial sums
__device__ __forceinline__ void bn_total(const double* __restrict__ part, int ch, int c, double& s1, double& s2, int ng = BN_G) {
  s1 = 0.0; s2 = 0.0;
  for (int g = 0; g < ng; ++g) { s1 += part[((size_t)g * ch + c) * 2]; s2 += part[((size_t)g * ch + c) * 2 + 1]; }
}
struct PairAffine { const float* mean_rstd; const float* gamma; const float* beta; int relu; };   // per channel; mean_rstd == nullptr: none
// forward side: the previous repeat only left the clip-group sums of its 1x1 output (chan_sums_kernel<0>); the depthwise kernel turns
// them into mean / rstd itself (a few double operations per wave), and the wave that owns clip 0 of a channel pair publishes mean_rstd
// for the backward pass and updates the running statistics -- no finalize launch, no device-scope fence (a fence writes back the
// whole XCD L2: 66 us per layer when tried)
struct PairBnIn {
  const double* part; const float* gamma; const float* beta; float eps; int relu; double n;
  float* mean_rstd; float* running_mean; float* running_var; float momentum; long long* nbt;
  const float* tiles = nullptr; int n_tiles = 0;          // instead of `part`: per-tile (sum, sum of squares) pairs, f32 [ch][n_tiles][2] (ts_tcs_desc.stats)
};
// the channel's totals out of the per-tile pairs: the 64 lanes of a wave share the tiles, then combine (every lane of the wave must call this)
__device__ __forceinline__ void bn_total_tiles(const float* __restrict__ tiles, int n_tiles, int ch, int c, int lane, double& s1, double& s2) {
  double a1 = 0.0, a2 = 0.0;
  const float* const row = tiles + (size_t)c * n_tiles * 2;
  for (int p = lane; p < n_tiles; p += 64) { a1 += (double)row[2 * p]; a2 += (double)row[2 * p + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
  s1 = a1; s2 = a2;
}

__device__ __forceinline__ RowAffine row_affine(const PairAffine& p, int c) {
  RowAffine r{1.f, 0.f, 1.f, 0.f, false, false};
  if (p.mean_rstd) {
    r.sa = p.gamma[c] * p.mean_rstd[2 * c + 1];     r.ha = p.beta[c] - p.mean_rstd[2 * c] * r.sa;
    r.sb = p.gamma[c + 1] * p.mean_rstd[2 * c + 3]; r.hb = p.beta[c + 1] - p.mean_rstd[2 * c + 2] * r.sb;
    r.relu = p.relu != 0; r.on = true;
  }
  return r;
}

__device__ __forceinline__ RowAffine row_affine(const PairBnIn& p, int c, int ch, bool publish) {
  RowAffine r{1.f, 0.f, 1.f, 0.f, false, false};
  if (p.part) {
    float sc[2], hs[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      double s1, s2;
      bn_total(p.part, ch, c + e, s1, s2);
      const double mu = s1 / p.n;
      double var = s2 / p.n - mu * mu;
      var = var < 0.0 ? 0.0 : var;
      const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
      sc[e] = p.gamma[c + e] * rstd; hs[e] = p.beta[c + e] - (float)mu * sc[e];
      if (publish) {
        p.mean_rstd[2 * (c + e)] = (float)mu; p.mean_rstd[2 * (c + e) + 1] = rstd;
        if (p.running_mean) {      // nn.BatchNorm1d's update: momentum blend of the batch mean and the UNBIASED batch variance
          p.running_mean[c + e] = (1.f - p.momentum) * p.running_mean[c + e] + p.momentum * (float)mu;
          p.running_var[c + e] = (1.f - p.momentum) * p.running_var[c + e] + p.momentum * (float)(var * (p.n / (p.n > 1.0 ? p.n - 1.0 : 1.0)));
          if (c + e == 0 && p.nbt) *p.nbt += 1;
        }
      }
    }
    r.sa = sc[0]; r.ha = hs[0]; r.sb = sc[1]; r.hb = hs[1]; r.relu = p.relu != 0; r.on = true;
  }
  return r;
}

template <class T, bool PH = false>
__device__ __forceinline__ void stage_write(const Staged<T>& st, v2f* dst, int q, int a0, int n, int lim, int lane,
                                            const RowAffine af = RowAffine{1.f, 0.f, 1.f, 0.f, false, false}) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int u = lane + 64 * it, i8 = a0 + 8 * u;
    if (8 * u >= n) continue;
    if (PH) {
      const int i16 = 2 * a0 + 16 * u;
      float f[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) f[m] = 0.f;
      if (i16 >= 0 && i16 < lim) {
        float lo[8], hi[8];
        raw_get(st.a[it], lo);
#pragma unroll
        for (int m = 0; m < 8; ++m) f[m] = i16 + m < lim ? lo[m] : 0.f;
        if (i16 + 8 < lim) {
          raw_get(st.b[it], hi);
#pragma unroll
          for (int m = 0; m < 8; ++m) f[8 + m] = i16 + 8 + m < lim ? hi[m] : 0.f;
        }
      }
      v2f* const d = dst + 2 * u;
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<v2fx2*>(d + r * q) = v2fx2{v2f{f[4 * r], f[4 * r + 1]}, v2f{f[4 * r + 2], f[4 * r + 3]}};
      continue;
    }
    float a[8], b[8];
    if (i8 >= 0 && i8 < lim) {
      raw_get(st.a[it], a); raw_get(st.b[it], b);
      if (af.on) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          a[m] = fmaf(a[m], af.sa, af.ha); b[m] = fmaf(b[m], af.sb, af.hb);
          if (af.relu) { a[m] = a[m] > 0.f ? a[m] : 0.f; b[m] = b[m] > 0.f ? b[m] : 0.f; }
        }
      }
      if (i8 + 8 > lim) {
#pragma unroll
        for (int m = 0; m < 8; ++m) if (i8 + m >= lim) { a[m] = 0.f; b[m] = 0.f; }
      }
    } else {
#pragma unroll
      for (int m = 0; m < 8; ++m) { a[m] = 0.f; b[m] = 0.f; }
    }
    v2f* const d = dst + 2 * u;
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<v2fx2*>(d + r * q) = v2fx2{v2f{a[2 * r], b[2 * r]}, v2f{a[2 * r + 1], b[2 * r + 1]}};
  }
}

// entries e0 .. e0 + 7 (e0 a multiple of 8) of a staged pair -> win[off .. off + 7]
__device__ __forceinline__ void win_load(const v2f* src, int q, int e0, v2f (&win)[16], int off) {
  const int base = (e0 >> 3) << 1;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const v2fx2 v = *reinterpret_cast<const v2fx2*>(src + base + r * q);
    win[off + 2 * r] = v.a; win[off + 2 * r + 1] = v.b;
  }
}

// One FIR over the pair: acc[m] += sum_j taps[j] * src[e0 + m + j], j < k8 (a multiple of 8; `taps` = float2 pairs in LDS, zero-padded,
// 16-byte aligned: every lane reads the same address, a broadcast).  The alternatives were measured: taps over the scalar memory path
// cost 1.5x the kernel time (SMEM and LDS share one wait counter, so every LDS wait also waits for the step's 16 scalar loads); taps
// in registers handed to the packed FMAs through v_readlane cost 1.13x (SGPR-write hazards, fewer waves).
__device__ __forceinline__ void fir_pair(const v2f* src, int q, int e0, int k8, const v2f* taps, v2f (&acc)[8]) {
  v2f win[16];
  win_load(src, q, e0, win, 0);
  for (int j0 = 0; j0 < k8; j0 += 8) {
    win_load(src, q, e0 + j0 + 8, win, 8);
    v2f w[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const v2fx2 v = *reinterpret_cast<const v2fx2*>(taps + j0 + 2 * r);
      w[2 * r] = v.a; w[2 * r + 1] = v.b;
    }
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[m] = __builtin_elementwise_fma(w[jj], win[m + jj], acc[m]);
#pragma unroll
    for (int m = 0; m < 8; ++m) win[m] = win[m + 8];
  }
}

__device__ __forceinline__ void store_pair(float* ra, float* rb, const v2f (&acc)[8], int t, int lim) {
  float a[8], b[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) { const bool on = t + m < lim; a[m] = on ? acc[m][0] : 0.f; b[m] = on ? acc[m][1] : 0.f; }
  store8(ra + t, a); store8(rb + t, b);
}
__device__ __forceinline__ void store_pair(bf16_t* ra, bf16_t* rb, const v2f (&acc)[8], int t, int lim) {
  float a[8], b[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) { const bool on = t + m < lim; a[m] = on ? acc[m][0] : 0.f; b[m] = on ? acc[m][1] : 0.f; }
  store8(ra + t, a); store8(rb + t, b);
}

// PH: entries te .. te + 7 of one row = frames 2 te .. 2 te + 15
template <class T>
__device__ __forceinline__ void store_phase(T* row, const v2f (&acc)[8], int te, int lim) {
  float a[8], b[8];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    a[2 * m] = 2 * te + 2 * m < lim ? acc[m][0] : 0.f;             a[2 * m + 1] = 2 * te + 2 * m + 1 < lim ? acc[m][1] : 0.f;
    b[2 * m] = 2 * te + 8 + 2 * m < lim ? acc[4 + m][0] : 0.f;     b[2 * m + 1] = 2 * te + 9 + 2 * m < lim ? acc[4 + m][1] : 0.f;
  }
  store8(row + 2 * te, a); store8(row + 2 * te + 8, b);
}

__host__ __device__ constexpr int pair_xl(int k8) { return PT + k8 + 16; }                       // staged x samples per tile
constexpr int PAIR_TAPS = DW_KMAX + 16;                                                          // float2 tap slots per wave (k + 7 rounded up to 8)
__host__ __device__ constexpr int pair_gl(int k, int p) { return round_up(p, 8) + PT + round_up(k + round_up(p, 8) - p, 8) + 16; }

// forward: y[r, t] = sum_j w[c, j] xm[r, t + j - p].  A wave walks over `pairs_per_wave` consecutive row pairs (times the 512-frame
// tiles of a row): while one pair is filtered out of LDS, the rows of the next are already on their way from HBM.
// PH: one wave item = ONE row of a dilation-2 layer as (even, odd) phases; `p` = padding / 2, tiles and FIR in entries (2 frames).
template <class T, bool PH = false>
__global__ __launch_bounds__(256) void dw_fwd_pair_kernel(const T* __restrict__ x, const int* __restrict__ len_in, const int* __restrict__ len_out,
                                                          const float* __restrict__ w, T* __restrict__ y, int batch, int ch, int t, int k, int p,
                                                          int pitch, int pairs_per_wave, PairBnIn aff) {
  extern __shared__ __attribute__((aligned(16))) v2f sm2[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int sh = (-p) & 7;                                 // (t0 - p) & 7 for every tile (PT is a multiple of 8): the taps move, not the samples
  const int k8 = round_up(k + sh, 8), xl = pair_xl(round_up(k + 7, 8)), q = xl >> 2;
  v2f* const xs = sm2 + wave * (xl + PAIR_TAPS);
  v2f* const tp = xs + xl;                                 // this wave's taps, shifted by sh and zero-padded
  constexpr int RPI = PH ? 1 : 2;                          // rows per item
  const int te = PH ? (t + 1) / 2 : t;                     // length in entries
  const long long n_pairs = (long long)batch * ch / RPI;
  const long long rp0 = ((long long)blockIdx.x * 4 + wave) * pairs_per_wave;
  if (rp0 >= n_pairs) return;
  const int np = rp0 + pairs_per_wave <= n_pairs ? pairs_per_wave : (int)(n_pairs - rp0);
  const int n_tiles = (te + PT - 1) / PT, items = np * n_tiles;
  Staged<T> st;
  auto fetch = [&](int it) {
    const long long rp = rp0 + it / n_tiles;
    const int b = (int)(rp * RPI / ch), t0 = (it % n_tiles) * PT;
    const T* const xa = x + (size_t)rp * RPI * pitch;
    stage_fetch<T, PH>(st, xa, xa + pitch, t0 - p - sh, xl, clamp_len(len_in, b, t), lane);
  };
  fetch(0);
  for (int it = 0; it < items; ++it) {
    const long long rp = rp0 + it / n_tiles;
    const int b = (int)(rp * RPI / ch), c = (int)(rp * RPI % ch), t0 = (it % n_tiles) * PT;
    const int li = clamp_len(len_in, b, t), lo = len_out ? clamp_len(len_out, b, t) : t;
    if (PH) stage_write<T, true>(st, xs, q, t0 - p - sh, xl, li, lane);
    else stage_write<T, false>(st, xs, q, t0 - p - sh, xl, li, lane, row_affine(aff, c, ch, b == 0 && t0 == 0 && lane == 0));
    // y[t0 + tt] = sum_j' w'[j'] xs[tt + j'],  xs[e] = xm[t0 - p - sh + e],  w'[j'] = w[j' - sh]
    if (it % n_tiles == 0) {
      const float* const wa = w + (size_t)c * k;
      const int wb = PH ? 0 : k;                           // phase split: both components use the row's own taps
      for (int j = lane; j < k8; j += 64) { const int jj = j - sh; tp[j] = (jj >= 0 && jj < k) ? v2f{wa[jj], wa[wb + jj]} : v2f{0.f, 0.f}; }
    }
    if (it + 1 < items) fetch(it + 1);
    __builtin_amdgcn_wave_barrier();
    T* const ya = y + (size_t)rp * RPI * pitch;
    const int t8 = 8 * lane;
    if (t0 + t8 < te) {
      v2f acc[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[m] = v2f{0.f, 0.f};
      fir_pair(xs + ((t8 >> 3) << 1), q, 0, k8, tp, acc);
      if (PH) store_phase(ya, acc, t0 + t8, lo); else store_pair(ya, ya + pitch, acc, t0 + t8, lo);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// Depthwise forward on the matrix cores (bf16 rows, >= 17 clips): for ONE channel the convolution of 32 clips is a matrix product,
//   Y[t, b] = sum_i Toep[t, i] X[i, b],   Toep[t, i] = w[i - t + p],
// v_mfma_f32_32x32x16_bf16 with M = 32 output frames, N = 32 CLIPS, K = 16 input frames.  A wave owns one channel, one group of 32
// clips and a run of output blocks, and streams over the input in chunks of 16 frames:
//   * B operand = chunk [16 frames][32 clips]: lane (clip, half) holds 8 consecutive frames of its clip = ONE 16-byte global load --
//     the input never goes through LDS (masking by length, and the previous repeat's BatchNorm + ReLU, act on the fragment in
//     registers); loads run DWM_AHEAD chunk pairs ahead through a register ring;
//   * A operand = a 32 x 16 slice of the Toeplitz matrix: chunk ic meets output block ob with relative index r = ic - 2 ob and
//     A_r[m, k'] = w[16 r + k' - m - sh] (sh = round_up(p, 16) - p: the chunks start 16-aligned at frame -round_up(p, 16)); the
//     R <= 2 NB fragments of the channel are built once per wave (taps rounded to bf16) and live in registers;
//   * block ob is met by chunks 2 ob .. 2 ob + R - 1, so NB = ceil(R / 2) accumulators are open at a time: acc[d] = block q - d while the
//     chunk pair q is processed; the oldest leaves after it, the others shift down;
//   * store: lane pairs (l, l ^ 32) exchange half their register groups so that each lane writes 8 consecutive frames (16 bytes).
// MFMA work: 2 NB instructions per 32 x 32 outputs (K63: 6, 63 / 94 of the products useful) against 64 packed FMAs per 8 x 2 outputs of
// the pair kernel; the kernel is bound by its memory traffic, not by arithmetic or LDS.
// ----------------------------------------------------------------------------------------------------------------------
#ifndef TS_DWM_WAVES_PER_CU
#define TS_DWM_WAVES_PER_CU 8      // time segments per (channel, clip group) are chosen for this many waves per CU (round-6 A/B of the C4 steps: 4 -> the same, 16 -> +4 %)
#endif
constexpr int DWM_AHEAD = 4;                             // chunk pairs in flight
constexpr int DWM_TAPG = 48, DWM_TAPN = 48 + 176;        // taps in LDS per wave: zero guards in front / behind

template <int NB>
__global__ __launch_bounds__(256) void dw_fwd_mfma_kernel(const bf16_t* __restrict__ x, const int* __restrict__ len_in, const int* __restrict__ len_out,
                                                          const float* __restrict__ w, bf16_t* __restrict__ y, int batch, int ch, int t, int k, int p,
                                                          int pitch, int n_seg, PairBnIn aff) {
  __shared__ float tapz[4][DWM_TAPN];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n_cg = (batch + 31) >> 5;
  const long long wid = (long long)blockIdx.x * 4 + wave;                 // wave id -> (channel, clip group, segment): segment fastest
  if (wid >= (long long)ch * n_cg * n_seg) return;
  const int seg = (int)(wid % n_seg), cg = (int)((wid / n_seg) % n_cg), c = (int)(wid / ((long long)n_seg * n_cg));
  const int n_blk = (t + 31) >> 5, per = (n_blk + n_seg - 1) / n_seg;
  const int ob_lo = seg * per, ob_hi = ob_lo + per < n_blk ? ob_lo + per : n_blk;
  if (ob_lo >= ob_hi) return;
  const int pup = round_up(p, 16), sh = pup - p, R = (k + 30 + sh) / 16 + 1;          // R <= 2 NB (launcher)
  const int m = lane & 31, h = lane >> 5;
  // ---- this lane's clip
  const int b = cg * 32 + m;
  const bool clip_on = b < batch;
  const int li = clip_on ? clamp_len(len_in, b, t) : 0, lo = clip_on ? (len_out ? clamp_len(len_out, b, t) : t) : 0;
  const bf16_t* const xr = x + ((size_t)(clip_on ? b : 0) * ch + c) * pitch;
  bf16_t* const yr = y + ((size_t)(clip_on ? b : 0) * ch + c) * pitch;
  // chunk pair q = chunks 2q (frames -pup + 32 q + 8 h ..) and 2q + 1 (+16): raw 16-byte loads, zero outside [0, li)
  auto fetch = [&](int q, u32x4& e, u32x4& o) {
    const int f0 = -pup + 32 * q + 8 * h, f1 = f0 + 16;
    e = (f0 >= 0 && f0 < li) ? *reinterpret_cast<const u32x4*>(xr + f0) : u32x4{0u, 0u, 0u, 0u};
    o = (f1 >= 0 && f1 < li) ? *reinterpret_cast<const u32x4*>(xr + f1) : u32x4{0u, 0u, 0u, 0u};
  };
  // the first chunk pairs are requested BEFORE the prologue below (tap fragments through LDS, the BatchNorm totals of 256 tiles): three dependent
  // round trips to memory in a 13-us launch become one
  const int q_lo = ob_lo, q_hi = ob_hi + NB - 1;           // block ob is complete after chunk pair ob + NB - 1
  u32x4 ring_e[DWM_AHEAD], ring_o[DWM_AHEAD];
#pragma unroll
  for (int a = 0; a < DWM_AHEAD; ++a) fetch(q_lo + a, ring_e[a], ring_o[a]);
  // the taps travel through registers so that their loads, too, are in flight before anything waits
  constexpr int NTV = (DWM_TAPN + 63) / 64;
  float tv[NTV];
#pragma unroll
  for (int i = 0; i < NTV; ++i) { const int j = lane + 64 * i - DWM_TAPG; tv[i] = (j >= 0 && j < k) ? w[(size_t)c * k + j] : 0.f; }
  // BatchNorm of the previous repeat on the fly (one channel: the pair helper on (c, c) would read c + 1 -> do it by hand)
  float sc = 1.f, hs = 0.f;
  bool aff_on = false, aff_relu = false;
  if (aff.part || aff.tiles) {
    double s1, s2;
    if (aff.tiles) bn_total_tiles(aff.tiles, aff.n_tiles, ch, c, lane, s1, s2); else bn_total(aff.part, ch, c, s1, s2);
    const double mu = s1 / aff.n;
    double var = s2 / aff.n - mu * mu;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)aff.eps));
    sc = aff.gamma[c] * rstd; hs = aff.beta[c] - (float)mu * sc;
    aff_on = true; aff_relu = aff.relu != 0;
    if (cg == 0 && seg == 0 && lane == 0) {
      aff.mean_rstd[2 * c] = (float)mu; aff.mean_rstd[2 * c + 1] = rstd;
      if (aff.running_mean) {
        aff.running_mean[c] = (1.f - aff.momentum) * aff.running_mean[c] + aff.momentum * (float)mu;
        aff.running_var[c] = (1.f - aff.momentum) * aff.running_var[c] + aff.momentum * (float)(var * (aff.n / (aff.n > 1.0 ? aff.n - 1.0 : 1.0)));
        if (c == 0 && aff.nbt) *aff.nbt += 1;
      }
    }
  }
  // ---- taps -> LDS (zero guards) -> the R Toeplitz fragments of this lane
  float* const tz = tapz[wave];
#pragma unroll
  for (int i = 0; i < NTV; ++i)
    if (lane + 64 * i < DWM_TAPN) tz[lane + 64 * i] = tv[i];
  __builtin_amdgcn_wave_barrier();
  s16x8 A[2 * NB];
#pragma unroll
  for (int r = 0; r < 2 * NB; ++r) {
    const float* const tp = tz + DWM_TAPG + 16 * r + 8 * h - m - sh;                // >= -46, <= 16 (2 NB - 1) + 15
    const u32x4 v = u32x4{pack_bf16(tp[0], tp[1]), pack_bf16(tp[2], tp[3]), pack_bf16(tp[4], tp[5]), pack_bf16(tp[6], tp[7])};
    A[r] = r < R ? __builtin_bit_cast(s16x8, v) : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  auto prep = [&](u32x4 v, int f0) {                      // input transform + length mask of one fragment
    if (f0 < 0 || f0 >= li) return s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (aff_on) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a0 = fmaf(bf16_lo(v[j]), sc, hs), a1 = fmaf(bf16_hi(v[j]), sc, hs);
        if (aff_relu) { a0 = a0 > 0.f ? a0 : 0.f; a1 = a1 > 0.f ? a1 : 0.f; }
        v[j] = pack_bf16(a0, a1);
      }
    }
    if (f0 + 8 > li) v = keep_first(v, li - f0);
    return __builtin_bit_cast(s16x8, v);
  };
  f32x16 acc[NB];
#pragma unroll
  for (int d = 0; d < NB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  for (int q0 = q_lo; q0 < q_hi; q0 += DWM_AHEAD) {
#pragma unroll
    for (int a = 0; a < DWM_AHEAD; ++a) {
      const int q = q0 + a;
      if (q < q_hi) {
        const s16x8 be = prep(ring_e[a], -pup + 32 * q + 8 * h), bo = prep(ring_o[a], -pup + 32 * q + 16 + 8 * h);
        fetch(q + DWM_AHEAD, ring_e[a], ring_o[a]);
#pragma unroll
        for (int d = 0; d < NB; ++d) {                     // acc[d] = block q - d: relative chunk indices 2 d (even chunk), 2 d + 1 (odd)
          acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2 * d], be, acc[d], 0, 0, 0);
          acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2 * d + 1], bo, acc[d], 0, 0, 0);
        }
        const int ob = q - (NB - 1);
        if (ob >= ob_lo && ob < ob_hi) {
          // register group rg of acc = frames 32 ob + 8 rg + 4 h + 0..3 of this lane's clip: pack, trade two groups with lane ^ 32,
          // store 8 consecutive frames twice
          unsigned pk[4][2];
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            pk[rg][0] = pack_bf16(acc[NB - 1][4 * rg + 0], acc[NB - 1][4 * rg + 1]);
            pk[rg][1] = pack_bf16(acc[NB - 1][4 * rg + 2], acc[NB - 1][4 * rg + 3]);
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            // group this lane finishes: 2 u + h; it sends the other one of the pair (2 u + 1 - h) to its partner
            const unsigned s0 = h ? pk[2 * u][0] : pk[2 * u + 1][0], s1 = h ? pk[2 * u][1] : pk[2 * u + 1][1];
            const unsigned g0 = __shfl_xor(s0, 32), g1 = __shfl_xor(s1, 32);
            const unsigned m0 = h ? pk[2 * u + 1][0] : pk[2 * u][0], m1 = h ? pk[2 * u + 1][1] : pk[2 * u][1];
            const int f = 32 * ob + 8 * (2 * u + h);
            u32x4 v = h ? u32x4{g0, g1, m0, m1} : u32x4{m0, m1, g0, g1};            // frames f .. f + 3 come from the h = 0 lane
            if (clip_on && f < pitch) {
              if (f + 8 > lo) v = keep_first(v, lo - f);
              st16(reinterpret_cast<u32x4*>(yr + f), v);
            }
          }
        }
#pragma unroll
        for (int d = NB - 1; d > 0; --d) acc[d] = acc[d - 1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
      }
    }
  }
}

// backward, data AND weights in one pass over dy / x:
//   dx[r, i] = (i < len_in) sum_j w[c, j] dym[r, i + p - j]          (the forward FIR with the taps flipped)
//   dw[c, j] += sum_{b, t} dym[r, t] xm[r, t + j - p]
// Workgroup = one channel pair x (4 waves x clips_per_wave clips); lane = (tap group of 8, frame slice) for the weight part,
// partial sums stay in registers over the wave's clips, are combined across the workgroup in LDS and leave as one atomicAdd per
// (channel, tap): dw ACCUMULATES (the caller hands in zeros for a plain gradient).
template <class T, bool PH = false>
__global__ __launch_bounds__(256) void dw_bwd_pair_kernel(const T* __restrict__ dy, const T* __restrict__ x, const int* __restrict__ len_in,
                                                          const int* __restrict__ len_out, const float* __restrict__ w, T* __restrict__ dx,
                                                          float* __restrict__ dw, int batch, int ch, int t, int k, int p, int pitch,
                                                          int clips_per_wave, PairAffine aff, float* __restrict__ in_dgamma,
                                                          float* __restrict__ in_dbeta, float* __restrict__ det_part) {
  extern __shared__ __attribute__((aligned(16))) v2f sm2[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int sh = (-p) & 7;                                 // x is staged from the 8-aligned frame t0 - p - sh: tap j sits at window offset j + sh
  const int k8 = round_up(k + sh, 8), xl = pair_xl(round_up(k + 7, 8)), qx = xl >> 2;
  const int o = round_up(p, 8), fpad = o - p, kf8 = round_up(k + fpad, 8), gl = pair_gl(k, p), qg = gl >> 2;
  v2f* const xs = sm2 + wave * (xl + gl + PAIR_TAPS);
  v2f* const gs = xs + xl;
  v2f* const tp = gs + gl;                                 // this wave's taps, flipped, padded by fpad zeros in front
  const int c = PH ? blockIdx.x : 2 * blockIdx.x;          // PH: one row per workgroup column, entries = (even, odd) frames, p = padding / 2
  const int te = PH ? (t + 1) / 2 : t;
  {
    const float* const wa = w + (size_t)c * k;
    const int wb = PH ? 0 : k;
    for (int j = lane; j < kf8; j += 64) { const int jj = j - fpad; tp[j] = (jj >= 0 && jj < k) ? v2f{wa[k - 1 - jj], wa[wb + k - 1 - jj]} : v2f{0.f, 0.f}; }
  }
  const int ng = k8 >> 3, nq = 64 / ng, g = lane % ng, sl = lane / ng;
  // dw == nullptr: the depthwise weight is frozen (first phase of the reference's fine-tuning schedule): data gradient only -- the x rows
  // are then not even staged (with an input transform the dx part reads the x values it needs straight from memory)
  const bool need_dw = dw != nullptr;
  const bool active = need_dw && sl < nq;
  v2f part[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) part[jj] = v2f{0.f, 0.f};
  // with an input transform (x = relu?(BatchNorm(v)) of the previous repeat, applied at staging) this kernel also does the first half of
  // that BatchNorm's backward: it stores g = dL/dy * (y > 0) instead of dL/dy and accumulates sum g (-> in_dbeta) and sum g * xhat
  // (-> in_dgamma) per channel -- exactly the two sums ts_train_bn_bwd_sums needs, and the parameter gradients themselves
  const RowAffine af = row_affine(aff, c);
  const float mu_a = aff.mean_rstd ? aff.mean_rstd[2 * c] : 0.f, rs_a = aff.mean_rstd ? aff.mean_rstd[2 * c + 1] : 0.f;
  const float mu_b = aff.mean_rstd ? aff.mean_rstd[2 * c + 2] : 0.f, rs_b = aff.mean_rstd ? aff.mean_rstd[2 * c + 3] : 0.f;
  v2f s1 = v2f{0.f, 0.f}, s2 = v2f{0.f, 0.f};
  const int b_lo = (blockIdx.y * 4 + wave) * clips_per_wave;
  const int nb = b_lo >= batch ? 0 : (b_lo + clips_per_wave <= batch ? clips_per_wave : batch - b_lo);
  const int n_tiles = (te + PT - 1) / PT, items = nb * n_tiles;
  Staged<T> sx, sg;
  auto fetch = [&](int it) {
    const int b = b_lo + it / n_tiles, t0 = (it % n_tiles) * PT;
    const size_t r0 = ((size_t)b * ch + c) * pitch;
    if (need_dw) stage_fetch<T, PH>(sx, x + r0, x + r0 + pitch, t0 - p - sh, xl, clamp_len(len_in, b, t), lane);
    stage_fetch<T, PH>(sg, dy + r0, dy + r0 + pitch, t0 - o, gl, len_out ? clamp_len(len_out, b, t) : t, lane);
  };
  if (items > 0) fetch(0);
  for (int it = 0; it < items; ++it) {
    const int b = b_lo + it / n_tiles, t0 = (it % n_tiles) * PT;
    const int li = clamp_len(len_in, b, t), lo = len_out ? clamp_len(len_out, b, t) : t;
    const size_t r0 = ((size_t)b * ch + c) * pitch;
    {
      if (need_dw) stage_write<T, PH>(sx, xs, qx, t0 - p - sh, xl, li, lane, af);
      stage_write<T, PH>(sg, gs, qg, t0 - o, gl, lo, lane);
      if (it + 1 < items) fetch(it + 1);                   // the next clip's rows travel while this one is filtered
      __builtin_amdgcn_wave_barrier();
      const int nt = te - t0 < PT ? te - t0 : PT;
      const int i8 = 8 * lane;
      if (i8 < nt) {
        v2f acc[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = v2f{0.f, 0.f};
        fir_pair(gs + ((i8 >> 3) << 1), qg, 0, kf8, tp, acc);
        if (!PH && af.on && t0 + i8 < li) {
          float va[8], vb[8];
          load8(x + r0 + t0 + i8, va); load8(x + r0 + pitch + t0 + i8, vb);
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const bool in = t0 + i8 + m < li;
            const float ya = fmaf(va[m], af.sa, af.ha), yb = fmaf(vb[m], af.sb, af.hb);
            const float ga = (in && (!af.relu || ya > 0.f)) ? acc[m][0] : 0.f, gb = (in && (!af.relu || yb > 0.f)) ? acc[m][1] : 0.f;
            acc[m] = v2f{ga, gb};
            s1 += acc[m];
            s2 += v2f{ga != 0.f ? ga * (va[m] - mu_a) * rs_a : 0.f, gb != 0.f ? gb * (vb[m] - mu_b) * rs_b : 0.f};   // pitch padding may hold NaN bits
          }
        }
        if (PH) store_phase(dx + r0, acc, t0 + i8, li); else store_pair(dx + r0, dx + r0 + pitch, acc, t0 + i8, li);
      }
      if (active) {
        const int per = round_up((nt + nq - 1) / nq, 8);
        const int lo_t = sl * per < nt ? sl * per : nt, hi_t = lo_t + per < nt ? lo_t + per : nt;
        if (lo_t < hi_t) {
          v2f win[16];
          win_load(xs, qx, lo_t + 8 * g, win, 0);
          for (int tt = lo_t; tt < hi_t; tt += 8) {
            win_load(xs, qx, tt + 8 * g + 8, win, 8);
            v2f gv[16];
            win_load(gs, qg, o + tt, gv, 0);
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
              for (int jj = 0; jj < 8; ++jj) part[jj] = __builtin_elementwise_fma(gv[m], win[m + jj], part[jj]);
#pragma unroll
            for (int m = 0; m < 8; ++m) win[m] = win[m + 8];
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  v2f* const red = sm2;                                  // [4 waves][64 lanes][8]
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) red[(threadIdx.x) * 8 + jj] = active ? part[jj] : v2f{0.f, 0.f};
  __syncthreads();
  for (int idx = threadIdx.x; need_dw && idx < (PH ? k : 2 * k); idx += 256) {
    const int sel = idx / k, j = idx % k, gj = (j + sh) >> 3, jj = (j + sh) & 7;      // register jj of tap group gj holds tap 8 gj + jj - sh
    float tot = 0.f;
    for (int wv = 0; wv < 4; ++wv)
      for (int r = 0; r < nq; ++r) {
        const v2f e = red[((wv * 64) + r * ng + gj) * 8 + jj];
        tot += PH ? e[0] + e[1] : e[sel];                  // phase split: both phases of the row feed the same tap
      }
    if (det_part) det_part[((size_t)blockIdx.y * ch + c + sel) * (k + 2) + j] = tot;
    else atomicAdd(dw + (size_t)(c + sel) * k + j, tot);
  }
  if (af.on) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s1[0] += __shfl_xor(s1[0], off); s1[1] += __shfl_xor(s1[1], off);
      s2[0] += __shfl_xor(s2[0], off); s2[1] += __shfl_xor(s2[1], off);
    }
    float* const rs = reinterpret_cast<float*>(sm2 + 2048);       // behind `red`: [4 waves][4]
    if (lane == 0) { rs[wave * 4] = s1[0]; rs[wave * 4 + 1] = s1[1]; rs[wave * 4 + 2] = s2[0]; rs[wave * 4 + 3] = s2[1]; }
    __syncthreads();
    if (threadIdx.x < 4) {
      const float tot = rs[threadIdx.x] + rs[4 + threadIdx.x] + rs[8 + threadIdx.x] + rs[12 + threadIdx.x];
      if (det_part) det_part[((size_t)blockIdx.y * ch + c + (threadIdx.x & 1)) * (k + 2) + k + (threadIdx.x < 2 ? 0 : 1)] = tot;
      else atomicAdd((threadIdx.x < 2 ? in_dbeta : in_dgamma) + c + (threadIdx.x & 1), tot);
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// Depthwise backward on the matrix cores -- autograd's backward of the depthwise MaskedConv1d of a training-mode block (reference
// quartznet/blocks.py:95-164 inside QuartznetBlock.forward, :317-338) -- for bf16 rows, "same" geometry, channels % 16 == 0: v_mfma_f32_4x4x4_16b_bf16 = 16 independent
// 4 x 4 x 4 products per instruction, the shape the inference kernel's depthwise producers use (csrc/tcs_split.hip).  A WAVE owns 16
// consecutive channels and walks units = (clip, TT-frame tile); per unit it stages the dy and x windows [t0 - H, t0 + TT + H) of its 16
// rows in wave-private LDS (zero outside the lengths; x through the previous repeat's BatchNorm + ReLU when that is folded in) and runs
//   * the WEIGHT gradient  dw[c][j] += sum_u dym[t0 + u] xm[t0 + u + j - p],  u in [0, TT):
//       block = (channel, group g of 16 taps), D[i][jj] += sum_k A[i][k] B[k][jj] with A[i][k] = dys[H + tau + k - i] (a 4-frame window of
//       dy, shifted by the lane's i: three dwords + v_alignbit), B[k][jj] = xs[H - p4 + tau + k + 4 jj + 16 g]  ->  tap' = i + 4 jj + 16 g,
//       tap = tap' - (p4 - p), p4 = round_up(p, 4).  B of (g, step s) is B of (g + 1, step s - 4): the windows slide through a register
//       queue, ONE new 8-byte LDS read per step serves all G groups.  The first / last step of a tile mask the elements u < 0 / u >= TT.
//   * the DATA gradient  dx[t0 + r] = sum_j w[c][j] dym[t0 + r + p - j]  as the inference producers' Toeplitz product: block = channel,
//       column = one of 4 runs of TT / 4 frames, A = rows of the (flipped) tap Toeplitz slices out of a two-copy tap image built once per
//       wave, B = 4-frame windows of dys sliding through registers; results leave through the (now free) x window as 16-byte row segments.
// With a folded BatchNorm the data gradient's epilogue is that BatchNorm's backward first half (g = dx * (y > 0), sum g, sum g * xhat), as in
// dw_bwd_pair_kernel.  Workgroup = 4 waves on the SAME 16 channels (different units): the weight gradient of the workgroup is combined in
// LDS and leaves as one atomic per (channel, tap).  The VALU pair kernel this replaces spent 45 us per 512-channel K63 layer at 32 x 501
// frames on 49 MB of traffic: FIR arithmetic, not memory.
// ----------------------------------------------------------------------------------------------------------------------
struct DwbArgs {
  const bf16_t* dy; const bf16_t* x; const int* len_in; const int* len_out; const float* w;
  bf16_t* dx; float* dw;
  int batch, ch, t, k, p, pitch, n_tiles, upw;
  PairAffine aff; float* in_dgamma; float* in_dbeta;
  float* part;        // deterministic mode (ts_train_set_deterministic): [gridDim.y][ch][k + 2] partials instead of atomics, summed in order by det_reduce_kernel
};
__host__ __device__ constexpr int dwb_pitch(int w_el, int mod_dw) {        // row pitch in elements: >= w_el, pitch / 2 == mod_dw (mod 64)
  const int dwords = (w_el + 1) / 2;
  return 2 * ((dwords + 63 - mod_dw) / 64 * 64 + mod_dw);
}

template <int TT, int NK, int G>
__global__ __launch_bounds__(256, TT <= 128 ? 2 : 1) void dw_bwd_mfma_kernel(const DwbArgs a) {
  constexpr int M = TT / 16, RUN = TT / 4, S = TT / 4;      // dx: M steps of 4 frames per run; dw: S + 1 steps of 4 frames
  constexpr int HMAX = 40;                                  // halo for K <= 75 (round_up(37, 8))
  constexpr int WY = TT + 2 * HMAX + 16, WX = TT + 2 * HMAX + 16;
  constexpr int RPY = dwb_pitch(WY, 4), RPX = dwb_pitch(WX, 8);
  constexpr int CST = (16 * NK + 16) % 32 == 16 ? 16 * NK + 16 : 16 * NK + 32;
  constexpr int WAVEB = 16 * RPY * 2 + 16 * RPX * 2;
  static_assert(16 * CST <= 16 * RPY * 2 && 16 * (4 * NK + 4) * 4 <= 16 * RPX * 2, "the tap image / tap rows borrow the windows' LDS");
  extern __shared__ __attribute__((aligned(16))) char dwb_smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  char* const dys = dwb_smem + (size_t)wave * WAVEB;
  char* const xs = dys + 16 * RPY * 2;
  char* const tapl = dys;                                   // the tap image lives in registers (T) before the first window is staged
  const int row = lane >> 2, q = lane & 3;                  // channel of the 16, and i / jj / run / staging sub-lane
  const int c = blockIdx.x * 16 + row;
  const int K = a.k, p = a.p, H = round_up(p, 8), p4 = round_up(p, 4), woff = H - p4, dl = p4 - p;
  const bool need_dw = a.dw != nullptr;
  // the first unit's rows are requested BEFORE the tap image is built (tap loads -> LDS -> fragments: two dependent round trips otherwise, in a launch of
  // 16-28 us); unit n + 1 is requested before the matrix work of unit n as before
  const int unit0 = (blockIdx.y * 4 + wave) * a.upw, n_units = a.batch * a.n_tiles;
  const int unit1 = unit0 + a.upw < n_units ? unit0 + a.upw : n_units;
  // staging: 16-byte chunks, lane (row, q) takes chunks q, q + 4, ... of its row; the loads of unit n + 1 are issued (into registers) before the
  // matrix work of unit n starts, the LDS writes follow when unit n is done
  const int wxn = H - p4 + TT + 16 * G + 4 > TT + 2 * H ? H - p4 + TT + 16 * G + 4 : TT + 2 * H;
  // (the data gradient's windows reach woff + TT + 4 NK frames into the dy rows whatever K is: a template NK larger than this K's k-steps meets
  // zero taps there, and what they multiply must be staged zeros, not stale LDS)
  const int wyn = woff + TT + 4 * NK > TT + 2 * H ? woff + TT + 4 * NK : TT + 2 * H;
  const int ncy = (wyn + 7) / 8, ncx = (wxn + 7) / 8;
  constexpr int NCQ = (WX / 8 + 3) / 4;
  u32x4 gy[NCQ], gx[NCQ];
  auto fetch = [&](int un) {
    const int b = un / a.n_tiles, t0 = (un % a.n_tiles) * TT;
    const int li = clamp_len(a.len_in, b, a.t), lo = a.len_out ? clamp_len(a.len_out, b, a.t) : a.t;
    const size_t r0 = ((size_t)b * a.ch + c) * a.pitch;
#pragma unroll
    for (int it = 0; it < NCQ; ++it) {
      const int ck = q + 4 * it, f = t0 - H + 8 * ck;
      gy[it] = (ck < ncy && f >= 0 && f < lo) ? *reinterpret_cast<const u32x4*>(a.dy + r0 + f) : u32x4{0u, 0u, 0u, 0u};
      gx[it] = (need_dw && ck < ncx && f >= 0 && f < li) ? *reinterpret_cast<const u32x4*>(a.x + r0 + f) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  if (unit0 < unit1) fetch(unit0);
  // ---- tap image of this wave's 16 channels: wp[x] = w[K + 2 + dl - x] for x in [3 + dl, K + 2 + dl], two copies (the second shifted by one
  // element), dwords interleaved: dword d of copy e at byte 8 d + 4 e
  {
    // the 16 tap rows first (coalesced, all loads in flight at once: 16 K floats = this wave's slice of w), flipped into wl[row][x] = wp[x]
    float* const wl = reinterpret_cast<float*>(xs);                        // [16][4 NK + 4], free until the first unit is staged
    constexpr int WLP = 4 * NK + 4, NWL = (16 * WLP + 63) / 64;
    const float* const wg = a.w + (size_t)blockIdx.x * 16 * K;
    float tv[NWL];
#pragma unroll
    for (int it = 0; it < NWL; ++it) {
      const int idx = it * 64 + lane, rr = idx / WLP, x = idx % WLP, j = K + 2 + dl - x;
      tv[it] = (rr < 16 && j >= 0 && j < K) ? wg[(size_t)rr * K + j] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NWL; ++it) { const int idx = it * 64 + lane; if (idx < 16 * WLP) wl[idx] = tv[it]; }
    __builtin_amdgcn_wave_barrier();
    const float* const wr = wl + row * WLP;
#pragma unroll
    for (int it = 0; it < (2 * NK + 2 + 3) / 4; ++it) {
      const int d = q + 4 * it;
      if (d <= 2 * NK + 1) {
        const float w0 = wr[2 * d], w1 = wr[2 * d + 1], w2 = 2 * d + 2 < WLP ? wr[2 * d + 2] : 0.f;
        unsigned* const o = reinterpret_cast<unsigned*>(tapl + row * CST + 8 * d);
        o[0] = pack_bf16(w0, w1);
        o[1] = pack_bf16(w1, w2);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  u32x2 T[NK];
  {
    const char* const trow = tapl + row * CST + ((lane & 1) ? 0 : 4) + ((lane & 3) < 2 ? 8 : 0);
#pragma unroll
    for (int kk = 0; kk < NK; ++kk)
      T[kk] = u32x2{*reinterpret_cast<const unsigned*>(trow + kk * 16), *reinterpret_cast<const unsigned*>(trow + kk * 16 + 8)};
  }
  __builtin_amdgcn_wave_barrier();
  // ---- folded BatchNorm of the input
  const bool af = a.aff.mean_rstd != nullptr;
  float sc = 1.f, hs = 0.f, mu = 0.f, rs = 0.f;
  if (af) { mu = a.aff.mean_rstd[2 * c]; rs = a.aff.mean_rstd[2 * c + 1]; sc = a.aff.gamma[c] * rs; hs = a.aff.beta[c] - mu * sc; }
  const bool relu = af && a.aff.relu != 0;
  float s1 = 0.f, s2 = 0.f;
  f32x4 E[G];
#pragma unroll
  for (int g = 0; g < G; ++g) E[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  // dw: masks of the first (u < 0) and last (u >= TT) step for this lane's shift i = q
  const unsigned mF0 = q == 0 ? ~0u : (q == 1 ? 0xFFFF0000u : 0u), mF1 = q == 3 ? 0xFFFF0000u : ~0u;
  const unsigned mL0 = q == 0 ? 0u : (q == 1 ? 0x0000FFFFu : ~0u), mL1 = q == 3 ? 0x0000FFFFu : 0u;
  const unsigned ash = (q & 1) ? 16u : 0u;
  const char* const arow = dys + (size_t)row * RPY * 2 + ((H - q - (q & 1)) >> 1) * 4;          // + 8 s per step
  const char* const brow = xs + (size_t)row * RPX * 2 + (H - p4 + 4 * q) * 2;                   // + 8 s' per queue entry
  const char* const xrow = dys + (size_t)row * RPY * 2 + (woff + q * RUN) * 2;                  // dx windows: + 8 u
  for (int un = unit0; un < unit1; ++un) {
    const int b = un / a.n_tiles, t0 = (un % a.n_tiles) * TT;
    const int li = clamp_len(a.len_in, b, a.t), lo = a.len_out ? clamp_len(a.len_out, b, a.t) : a.t;
    const size_t r0 = ((size_t)b * a.ch + c) * a.pitch;
#pragma unroll
    for (int it = 0; it < NCQ; ++it) {
      const int ck = q + 4 * it, f = t0 - H + 8 * ck;
      if (ck < ncy) {
        u32x4 v = gy[it];
        if (f < lo && f + 8 > lo) v = keep_first(v, lo - f);
        *reinterpret_cast<u32x4*>(dys + (size_t)row * RPY * 2 + 16 * ck) = v;
      }
      if (need_dw && ck < ncx) {
        u32x4 v = gx[it];
        if (f >= 0 && f < li) {
          if (af) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float y0 = fmaf(bf16_lo(v[j]), sc, hs), y1 = fmaf(bf16_hi(v[j]), sc, hs);
              if (relu) { y0 = y0 > 0.f ? y0 : 0.f; y1 = y1 > 0.f ? y1 : 0.f; }
              v[j] = pack_bf16(y0, y1);
            }
          }
          if (f + 8 > li) v = keep_first(v, li - f);
        }
        *reinterpret_cast<u32x4*>(xs + (size_t)row * RPX * 2 + 16 * ck) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (un + 1 < unit1) fetch(un + 1);
    // the raw input under this tile's output chunks, for the folded BatchNorm's backward in the epilogue: requested now, used after the products
    constexpr int NOQ = TT / 32;
    u32x4 vraw[NOQ];
    if (af) {
#pragma unroll
      for (int it = 0; it < NOQ; ++it) {
        const int f = t0 + 8 * (q + 4 * it);
        vraw[it] = f < li ? *reinterpret_cast<const u32x4*>(a.x + r0 + f) : u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int it = 0; it < NOQ; ++it) {                   // zero from the length on: the sums below then need no per-element guard
        const int f = t0 + 8 * (q + 4 * it);
        if (f < li && f + 8 > li) vraw[it] = keep_first(vraw[it], li - f);
      }
    }
    // ---- weight gradient
    if (need_dw) {
      constexpr int QL = 4 * G;
      s16x4 R[QL];
#pragma unroll
      for (int u = 0; u < 4 * (G - 1); ++u) R[u] = *reinterpret_cast<const s16x4*>(brow + 8 * u);
#pragma unroll
      for (int s_ = 0; s_ <= S; ++s_) {
        R[(s_ + 4 * (G - 1)) % QL] = *reinterpret_cast<const s16x4*>(brow + 8 * (s_ + 4 * (G - 1)));
        const unsigned* const ap = reinterpret_cast<const unsigned*>(arow + 8 * s_);
        const unsigned d0 = ap[0], d1 = ap[1], d2 = ap[2];
        unsigned a0 = __builtin_amdgcn_alignbit(d1, d0, ash), a1 = __builtin_amdgcn_alignbit(d2, d1, ash);
        if (s_ == 0) { a0 &= mF0; a1 &= mF1; }
        if (s_ == S) { a0 &= mL0; a1 &= mL1; }
        const s16x4 av = __builtin_bit_cast(s16x4, u32x2{a0, a1});
#pragma unroll
        for (int g = 0; g < G; ++g) E[g] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(av, R[(s_ + 4 * g) % QL], E[g], 0, 0, 0);
      }
    }
    // ---- data gradient
    f32x4 d[M];
    {
      s16x4 P[NK + M - 1];
#pragma unroll
      for (int u = 0; u < NK + M - 1; ++u) P[u] = *reinterpret_cast<const s16x4*>(xrow + 8 * u);
#pragma unroll
      for (int kk = 0; kk < NK; ++kk)
#pragma unroll
        for (int m = 0; m < M; ++m)
          d[m] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, T[kk]), P[kk + m], kk == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : d[m], 0, 0, 0);
    }
    // results -> the x window's LDS (free now) as bf16 [16 rows][TT], then out in 16-byte row segments
    __builtin_amdgcn_wave_barrier();
    char* const ot = xs;                                   // row pitch TT * 2 + 16 bytes
    constexpr int OP = TT * 2 + 16;
#pragma unroll
    for (int m = 0; m < M; ++m)
      *reinterpret_cast<u32x2*>(ot + (size_t)row * OP + (q * RUN + 4 * m) * 2) = u32x2{pack_bf16(d[m][0], d[m][1]), pack_bf16(d[m][2], d[m][3])};
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NOQ; ++it) {
      const int ck = q + 4 * it, f = t0 + 8 * ck;
      if (f >= a.pitch) break;
      u32x4 v = *reinterpret_cast<const u32x4*>(ot + (size_t)row * OP + 16 * ck);
      if (f >= li) v = u32x4{0u, 0u, 0u, 0u}; else if (f + 8 > li) v = keep_first(v, li - f);
      if (af && f < li) {
        // g = dx * (y > 0); s1 = sum g; s2 accumulates sum g * v here -- sum g * xhat = rstd * (sum g v - mean * sum g) is formed once per channel
        const u32x4 vv = vraw[it];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v0 = bf16_lo(vv[j]), v1 = bf16_hi(vv[j]);
          float g0 = bf16_lo(v[j]), g1 = bf16_hi(v[j]);
          if (relu) { g0 = fmaf(v0, sc, hs) > 0.f ? g0 : 0.f; g1 = fmaf(v1, sc, hs) > 0.f ? g1 : 0.f; }
          s1 += g0 + g1;
          s2 = fmaf(g0, v0, fmaf(g1, v1, s2));
          if (relu) v[j] = pack_bf16(g0, g1);
        }
      }
      st16(reinterpret_cast<u32x4*>(a.dx + r0 + f), v);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // ---- flush: folded-BatchNorm sums (4 lanes per channel), then the workgroup's weight gradient
  if (af) {
    // one atomic per channel and WORKGROUP: 64 waves adding to the same address one by one cost 8 us per layer
    s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);
    s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
    s2 = (s2 - mu * s1) * rs;
    __syncthreads();
    float* const rb = reinterpret_cast<float*>(dwb_smem) + 4 * 16 * 16 * G;     // behind the weight-gradient partials: [4 waves][16][2]
    if (q == 0) { rb[(wave * 16 + row) * 2] = s1; rb[(wave * 16 + row) * 2 + 1] = s2; }
    __syncthreads();
    if (threadIdx.x < 32) {
      const int rr = threadIdx.x >> 1, e = threadIdx.x & 1;
      const float tot = rb[(0 * 16 + rr) * 2 + e] + rb[(1 * 16 + rr) * 2 + e] + rb[(2 * 16 + rr) * 2 + e] + rb[(3 * 16 + rr) * 2 + e];
      if (a.part) a.part[((size_t)blockIdx.y * a.ch + blockIdx.x * 16 + rr) * (K + 2) + K + e] = tot;
      else atomicAdd((e ? a.in_dgamma : a.in_dbeta) + blockIdx.x * 16 + rr, tot);
    }
  }
  if (need_dw) {
    __syncthreads();
    float* const red = reinterpret_cast<float*>(dwb_smem);                  // [4 waves][16 channels][16 G taps']
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[((size_t)wave * 16 + row) * (16 * G) + 16 * g + 4 * q + i] = E[g][i];
    __syncthreads();
    for (int idx = threadIdx.x; idx < 16 * K; idx += 256) {
      const int rr = idx / K, j = idx % K, tp = j + dl;
      float tot = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) tot += red[((size_t)wv * 16 + rr) * (16 * G) + tp];
      if (a.part) a.part[((size_t)blockIdx.y * a.ch + blockIdx.x * 16 + rr) * (K + 2) + j] = tot;
      else atomicAdd(a.dw + (size_t)(blockIdx.x * 16 + rr) * K + j, tot);
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// Row-wise streaming kernels: one WAVE = one (row, 512-frame chunk) unit, 64 lanes x 8 elements (16 / 32 bytes per lane), four
// units per 256-thread workgroup (a 10 s clip is 501 frames: with one workgroup per row three of its four waves had nothing to do).
// Rows are pitched and 16-byte aligned, so every access is a whole vector; columns >= t are scratch and may be overwritten.
// ----------------------------------------------------------------------------------------------------------------------
constexpr int ROW_CHUNK = 512;
// defines `row`, `chunk`, `i` (first frame of this lane) and returns from the kernel when the unit lies outside the tensor
#define TS_ROW_UNIT(n_rows)                                                                   \
  const int _cpr = (t + ROW_CHUNK - 1) / ROW_CHUNK;                                           \
  const long long _u = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);                        \
  const long long _r = _u / _cpr;                                                             \
  const int row = (int)_r, chunk = (int)(_u % _cpr), i = chunk * ROW_CHUNK + (threadIdx.x & 63) * 8; \
  if (_r >= (long long)(n_rows) || i >= t) return

// y = x with frames >= len[b] zeroed (the re-masking in front of every MaskedConv1d, and of gradients on the way back)
template <class T>
__global__ __launch_bounds__(256) void mask_time_kernel(const T* __restrict__ x, const int* __restrict__ len, T* __restrict__ y,
                                                        int batch, int ch, int t, int pitch_x, int pitch_y) {
  TS_ROW_UNIT((long long)batch * ch);
  const int b = row / ch;
  const int l = clamp_len(len, b, t);
  float v[8];
  load8(x + (size_t)row * pitch_x + i, v);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = i + j < l ? v[j] : 0.f;
  store8(y + (size_t)row * pitch_y + i, v);
}

// Per-channel sums over all B*T frames in BN_G clip groups (grid ch x BN_G): part[g][c] = (s1, s2), fp64 accumulation; the
// consumers add the BN_G partials in a fixed order (deterministic, no atomics).
//   MODE 0 (forward statistics):  s1 = sum v,  s2 = sum v^2
//   MODE 1 (backward statistics): g = dy * (y > 0 if relu), xhat = (v - mean) * rstd:  s1 = sum g,  s2 = sum g * xhat
//          -- g and xhat are recomputed here and in the apply kernel instead of being written out and read back twice
template <int MODE, class T>
__global__ __launch_bounds__(256) void chan_sums_kernel(const T* __restrict__ a, const T* __restrict__ y, const T* __restrict__ v,
                                                        const float* __restrict__ mean_rstd, double* __restrict__ part, int batch,
                                                        int ch, int t, int pitch, int relu) {
  __shared__ double r1[256], r2[256];
  const int c = blockIdx.x, grp = blockIdx.y;
  const int per = (batch + BN_G - 1) / BN_G;
  const int b_lo = grp * per, b_hi = b_lo + per < batch ? b_lo + per : batch;
  float mu = 0.f, rs = 0.f;
  if (MODE == 1) { mu = mean_rstd[2 * c]; rs = mean_rstd[2 * c + 1]; }
  double s1 = 0.0, s2 = 0.0;
  for (int b = b_lo + (threadIdx.x >> 6); b < b_hi; b += 4) {          // one wave per clip row (a 10 s clip = 501 frames = 63 lanes x 8)
    const size_t row = ((size_t)b * ch + c) * pitch;
    for (int i = (threadIdx.x & 63) * 8; i < t; i += 512) {
      float va[8], vy[8], vv[8];
      load8(a + row + i, va);
      if (MODE == 1) { load8(v + row + i, vv); if (relu) load8(y + row + i, vy); }
      float p1 = 0.f, p2 = 0.f;                 // 8 terms in f32, then f64 across the row
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (i + j < t) {
          if (MODE == 0) { p1 += va[j]; p2 = fmaf(va[j], va[j], p2); }
          else {
            const float gv = (relu && !(vy[j] > 0.f)) ? 0.f : va[j];
            p1 += gv;
            p2 = fmaf(gv, (vv[j] - mu) * rs, p2);
          }
        }
      }
      s1 += (double)p1; s2 += (double)p2;
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

// BatchNorm(train) forward: stats[c] = (sum v, sum v^2) -> mean, rstd (biased variance, eps), y = gamma*(v-mean)*rstd + beta [ReLU].
template <class T>
__global__ __launch_bounds__(256) void bn_fwd_kernel(const T* __restrict__ v, const double* __restrict__ part,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     T* __restrict__ y, float* __restrict__ mean_rstd, int batch, int ch, int t, int pitch,
                                                     float eps, int relu, float* __restrict__ running_mean,
                                                     float* __restrict__ running_var, float momentum,
                                                     long long* __restrict__ num_batches_tracked, int ng) {
  TS_ROW_UNIT((long long)batch * ch);
  const int c = row % ch;
  float mu, sc;
  {                                                        // every lane forms the channel's statistics (a few double operations)
    const double n = (double)batch * t;
    double s1, s2;
    bn_total(part, ch, c, s1, s2, ng);
    const double m = s1 / n;
    double var = s2 / n - m * m;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    mu = (float)m; sc = gamma[c] * rstd;
    if (row < ch && chunk == 0 && (threadIdx.x & 63) == 0) {     // clip 0's wave of this channel publishes the statistics
      mean_rstd[2 * c] = mu; mean_rstd[2 * c + 1] = rstd;
      if (running_mean) {    // nn.BatchNorm1d's update: momentum blend of the batch mean and the UNBIASED batch variance
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * (n / (n > 1.0 ? n - 1.0 : 1.0)));
        if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
      }
    }
  }
  const float be = beta[c];
  float x[8];
  load8(v + (size_t)row * pitch + i, x);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float o = sc * (x[j] - mu) + be;
    x[j] = (relu && !(o > 0.f)) ? 0.f : o;
  }
  store8(y + (size_t)row * pitch + i, x);
}

// Block tail (quartznet/blocks.py:332-337): out = relu(BatchNorm(v_a) + BatchNorm(v_b)) -- main branch and residual branch -- from the
// clip-group sums of both in ONE pass (instead of two BatchNorm apply passes and an add + ReLU pass); publishes both mean_rstd and
// applies both running-statistics updates.
struct BnSide {
  const void* v; const double* part; const float* gamma; const float* beta; float eps;
  float* mean_rstd; float* running_mean; float* running_var; float momentum; long long* nbt;
};
template <class T>
__global__ __launch_bounds__(256) void bn2_add_relu_kernel(BnSide a, BnSide b, T* __restrict__ out, int batch, int ch, int t, int pitch) {
  TS_ROW_UNIT((long long)batch * ch);
  const int c = row % ch;
  float scs[2], hs[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const BnSide& s = e == 0 ? a : b;
    const double n = (double)batch * t;
    double s1, s2;
    bn_total(s.part, ch, c, s1, s2);
    const double mu = s1 / n;
    double var = s2 / n - mu * mu;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)s.eps));
    scs[e] = s.gamma[c] * rstd; hs[e] = s.beta[c] - (float)mu * scs[e];
    if (row < ch && chunk == 0 && (threadIdx.x & 63) == 0) {
      s.mean_rstd[2 * c] = (float)mu; s.mean_rstd[2 * c + 1] = rstd;
      if (s.running_mean) {
        s.running_mean[c] = (1.f - s.momentum) * s.running_mean[c] + s.momentum * (float)mu;
        s.running_var[c] = (1.f - s.momentum) * s.running_var[c] + s.momentum * (float)(var * (n / (n > 1.0 ? n - 1.0 : 1.0)));
        if (c == 0 && s.nbt) *s.nbt += 1;
      }
    }
  }
  const float sa = scs[0], ha = hs[0] + hs[1], sb = scs[1];
  float x[8], z[8];
  load8(static_cast<const T*>(a.v) + (size_t)row * pitch + i, x);
  load8(static_cast<const T*>(b.v) + (size_t)row * pitch + i, z);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float o = fmaf(x[j], sa, fmaf(z[j], sb, ha));
    x[j] = o > 0.f ? o : 0.f;
  }
  store8(out + (size_t)row * pitch + i, x);
}

// ----------------------------------------------------------------------------------------------------------------------
// Block tail in ONE launch each way (round 5): a workgroup owns a CHANNEL -- its batch x ceil(t / 512) row units of both branches live in
// registers between the statistics and the apply step, so every tensor is read once and written once and no second launch has to wait for the
// channel sums.  Replaces, per block, chan_sums<0> x 2 + bn2_add_relu (forward: 23 us -> one launch) and (chan_sums<1> + bn_bwd_apply) x 2
// (backward: 38 us -> one launch) at 32 x 501 frames.  A wave holds up to CU_MAX units; larger batches keep the two-step kernels.
// The variance is formed around the mean (two passes over the registers), not as E[x^2] - mean^2: f32 is then enough per lane, the sums across
// lanes and waves run in f64 like the clip-group partials they replace.
// ----------------------------------------------------------------------------------------------------------------------
// rows stay in registers in their STORAGE form (bf16 rows: 4 VGPRs per 8 frames) and are widened where they are used
template <class T> struct ChanRegs;
template <> struct ChanRegs<bf16_t> {
  static constexpr int UMAX = 8;
  typedef u32x4 raw;
  static __device__ __forceinline__ raw load(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
  static __device__ __forceinline__ void widen(const raw& r, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = bf16_lo(r[j]); v[2 * j + 1] = bf16_hi(r[j]); }
  }
  static __device__ __forceinline__ raw narrow(const float (&v)[8]) {        // exact for values that are bf16 already
    return u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
  }
  // the compiler would rather keep the widened floats alive across the reduction than convert twice (256 VGPRs, one wave per SIMD): an opaque
  // touch of the storage registers makes the second widening a new computation
  static __device__ __forceinline__ void pin(raw& r) { asm volatile("" : "+v"(r)); }
};
template <> struct ChanRegs<float> {
  static constexpr int UMAX = 4;
  struct raw { f32x4 lo, hi; };
  static __device__ __forceinline__ raw load(const float* p) { return raw{*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4)}; }
  static __device__ __forceinline__ void widen(const raw& r, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = r.lo[j]; v[4 + j] = r.hi[j]; }
  }
  static __device__ __forceinline__ raw narrow(const float (&v)[8]) { return raw{f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}}; }
  static __device__ __forceinline__ void pin(raw&) {}
};

__device__ __forceinline__ double chan_reduce(double v, double* red) {      // all 256 threads -> the sum, in every thread
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();                                                          // `red` may still be read from the previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

struct Bn2FwdArgs {
  BnSide a, b;
  void* out;
  int batch, ch, t, pitch;
};

template <class T>
__global__ __launch_bounds__(256) void bn2_fwd_chan_kernel(const Bn2FwdArgs g) {
  typedef ChanRegs<T> R;
  constexpr int UMAX = R::UMAX;
  __shared__ double red[4];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cpr = (g.t + ROW_CHUNK - 1) / ROW_CHUNK, units = g.batch * cpr;
  const T* const va = static_cast<const T*>(g.a.v);
  const T* const vb = static_cast<const T*>(g.b.v);
  typename R::raw xa[UMAX], xb[UMAX];
  int off[UMAX], nval[UMAX];                       // element offsets fit 31 bits (checked by the launcher)
#pragma unroll
  for (int u = 0; u < UMAX; ++u) {
    const int unit = wave + 4 * u;
    const int b = unit / cpr, i = (unit % cpr) * ROW_CHUNK + lane * 8;
    nval[u] = unit < units ? (g.t - i < 0 ? 0 : (g.t - i > 8 ? 8 : g.t - i)) : 0;
    off[u] = nval[u] > 0 ? (b * g.ch + c) * g.pitch + i : 0;
    xa[u] = R::load(va + off[u]);                  // idle lanes re-read element 0: in bounds, never used
    xb[u] = R::load(vb + off[u]);
  }
  const double n = (double)g.batch * g.t;
  float scs[2], hs[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const BnSide& sd = e == 0 ? g.a : g.b;
    float p = 0.f;
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      float x[8];
      R::widen(e == 0 ? xa[u] : xb[u], x);
#pragma unroll
      for (int j = 0; j < 8; ++j) p += j < nval[u] ? x[j] : 0.f;
    }
    const double mu = chan_reduce((double)p, red) / n;
    const float mf = (float)mu;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      float x[8];
      R::pin(e == 0 ? xa[u] : xb[u]);
      R::widen(e == 0 ? xa[u] : xb[u], x);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = x[j] - mf;
        q = j < nval[u] ? fmaf(d, d, q) : q;
      }
    }
    double var = chan_reduce((double)q, red) / n;
    var -= (mu - (double)mf) * (mu - (double)mf);              // the deviations were taken from the f32-rounded mean
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)sd.eps));
    scs[e] = sd.gamma[c] * rstd; hs[e] = sd.beta[c] - mf * scs[e];
    if (threadIdx.x == 0) {
      sd.mean_rstd[2 * c] = mf; sd.mean_rstd[2 * c + 1] = rstd;
      if (sd.running_mean) {
        sd.running_mean[c] = (1.f - sd.momentum) * sd.running_mean[c] + sd.momentum * mf;
        sd.running_var[c] = (1.f - sd.momentum) * sd.running_var[c] + sd.momentum * (float)(var * (n / (n > 1.0 ? n - 1.0 : 1.0)));
        if (c == 0 && sd.nbt) *sd.nbt += 1;
      }
    }
  }
  const float sa = scs[0], sb = scs[1], ha = hs[0] + hs[1];
  T* const out = static_cast<T*>(g.out);
#pragma unroll
  for (int u = 0; u < UMAX; ++u) {
    if (nval[u] > 0) {
      float x[8], z[8], o[8];
      R::pin(xa[u]);
      R::pin(xb[u]);
      R::widen(xa[u], x);
      R::widen(xb[u], z);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float r = fmaf(x[j], sa, fmaf(z[j], sb, ha));
        o[j] = r > 0.f ? r : 0.f;
      }
      store8(out + off[u], o);
    }
  }
}

struct Bn2BwdArgs {
  const void* dout; const void* dout2; const int* len2;     // dout2 (may be NULL): a second gradient of `out`, counted for frames < len2[clip] only
  const void* out; const void* va; const void* vb;
  const float* gamma_a; const float* mr_a; const float* gamma_b; const float* mr_b;
  void* dva; void* dvb;
  float* dgamma_a; float* dbeta_a; float* dgamma_b; float* dbeta_b;
  int batch, ch, t, pitch;
};

// out = relu(BN_a(va) + BN_b(vb)):  g = dout * (out > 0);  dv_e = gamma_e rstd_e (g - mean(g) - xhat_e mean(g xhat_e)),  dbeta_e = sum g, dgamma_e = sum g xhat_e
template <class T>
__global__ __launch_bounds__(256) void bn2_bwd_chan_kernel(const Bn2BwdArgs g) {
  typedef ChanRegs<T> R;
  constexpr int UMAX = R::UMAX;
  __shared__ double red[4];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cpr = (g.t + ROW_CHUNK - 1) / ROW_CHUNK, units = g.batch * cpr;
  const float mua = g.mr_a[2 * c], rsa = g.mr_a[2 * c + 1], mub = g.mr_b[2 * c], rsb = g.mr_b[2 * c + 1];
  typename R::raw gg[UMAX], xa[UMAX], xb[UMAX];    // the gated gradient and both un-normalised inputs, in storage form
  int off[UMAX], nval[UMAX];
  float p0 = 0.f, pa = 0.f, pb = 0.f;
  // loads in batches of UB units (left alone the scheduler hoists all 4 UMAX row loads to the top: 256 VGPRs, one wave per SIMD)
  constexpr int UB = UMAX < 4 ? UMAX : 4;
#pragma unroll
  for (int u0 = 0; u0 < UMAX; u0 += UB) {
    typename R::raw dr[UB], orr[UB];
#pragma unroll
    for (int k = 0; k < UB; ++k) {
      const int u = u0 + k, unit = wave + 4 * u;
      const int b = unit / cpr, i = (unit % cpr) * ROW_CHUNK + lane * 8;
      nval[u] = unit < units ? (g.t - i < 0 ? 0 : (g.t - i > 8 ? 8 : g.t - i)) : 0;
      off[u] = nval[u] > 0 ? (b * g.ch + c) * g.pitch + i : 0;
      xa[u] = R::load(static_cast<const T*>(g.va) + off[u]);
      xb[u] = R::load(static_cast<const T*>(g.vb) + off[u]);
      dr[k] = R::load(static_cast<const T*>(g.dout) + off[u]);
      orr[k] = R::load(static_cast<const T*>(g.out) + off[u]);
      if (g.dout2) {
        // the block's output fed two consumers (the next block's main and residual branch): their gradients are added HERE instead of in a pass
        // of their own (Fork.backward's ts_train_add); the residual branch's input mask zeroes its share from the clip's length on
        float d1[8], d2[8];
        R::widen(dr[k], d1);
        R::widen(R::load(static_cast<const T*>(g.dout2) + off[u]), d2);
        const int l2 = g.len2 ? g.len2[b] : 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 8; ++j) d1[j] += i + j < l2 ? d2[j] : 0.f;
        dr[k] = R::narrow(d1);                     // bf16 rows: rounded like the stored sum of the separate pass
      }
    }
#pragma unroll
    for (int k = 0; k < UB; ++k) {
      const int u = u0 + k;
      float d[8], o[8], x[8], z[8];
      R::widen(dr[k], d);
      R::widen(orr[k], o);
      R::widen(xa[u], x);
      R::widen(xb[u], z);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool in = j < nval[u];                       // columns >= t are scratch: they may hold anything, NaN included
        d[j] = (in && o[j] > 0.f) ? d[j] : 0.f;
        p0 += d[j];
        pa = fmaf(d[j], in ? (x[j] - mua) * rsa : 0.f, pa);
        pb = fmaf(d[j], in ? (z[j] - mub) * rsb : 0.f, pb);
      }
      gg[u] = R::narrow(d);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const double n = (double)g.batch * g.t;
  const double s0 = chan_reduce((double)p0, red), sa = chan_reduce((double)pa, red), sb = chan_reduce((double)pb, red);
  if (threadIdx.x == 0) {
    g.dbeta_a[c] = (float)s0; g.dgamma_a[c] = (float)sa;
    g.dbeta_b[c] = (float)s0; g.dgamma_b[c] = (float)sb;
  }
  const float mg = (float)(s0 / n), mga = (float)(sa / n), mgb = (float)(sb / n);
  const float ka = g.gamma_a[c] * rsa, kb = g.gamma_b[c] * rsb;
#pragma unroll
  for (int u = 0; u < UMAX; ++u) {
    if (nval[u] > 0) {
      float d[8], x[8], z[8], da[8], db[8];
      R::pin(gg[u]);
      R::pin(xa[u]);
      R::pin(xb[u]);
      R::widen(gg[u], d);
      R::widen(xa[u], x);
      R::widen(xb[u], z);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        da[j] = ka * (d[j] - mg - (x[j] - mua) * rsa * mga);
        db[j] = kb * (d[j] - mg - (z[j] - mub) * rsb * mgb);
      }
      store8(static_cast<T*>(g.dva) + off[u], da);
      store8(static_cast<T*>(g.dvb) + off[u], db);
    }
  }
}

// dv = gamma*rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy * (y > 0) when relu,  xhat = (v - mean) * rstd
template <class T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ v,
                                                           const double* __restrict__ part, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean_rstd, T* __restrict__ dv,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int batch, int ch, int t,
                                                           int pitch, int relu) {
  TS_ROW_UNIT((long long)batch * ch);
  const int c = row % ch;
  float mg, mgx;
  {
    const double n = (double)batch * t;
    double s1, s2;
    bn_total(part, ch, c, s1, s2);
    mg = (float)(s1 / n); mgx = (float)(s2 / n);
    if (row < ch && chunk == 0 && (threadIdx.x & 63) == 0) { dbeta[c] = (float)s1; dgamma[c] = (float)s2; }
  }
  const float mu = mean_rstd[2 * c], rs = mean_rstd[2 * c + 1], k = gamma[c] * rs;
  const size_t base = (size_t)row * pitch + i;
  float g[8], vy[8], vv[8];
  load8(dy + base, g);
  load8(v + base, vv);
  if (relu) load8(y + base, vy);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float gj = (relu && !(vy[j] > 0.f)) ? 0.f : g[j];
    g[j] = k * (gj - mg - (vv[j] - mu) * rs * mgx);
  }
  store8(dv + base, g);
}

// second half of a BatchNorm(train) backward whose first half ran in dw_bwd_pair_kernel's epilogue: g = dL/dy * (y > 0) is stored,
// S1 = sum g (= dbeta) and S2 = sum g * xhat (= dgamma) are complete:  dv = gamma * rstd * (g - S1 / n - xhat * S2 / n)
template <class T>
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const T* __restrict__ g, const T* __restrict__ v, const float* __restrict__ gamma,
                                                          const float* __restrict__ mean_rstd, const float* __restrict__ dgamma,
                                                          const float* __restrict__ dbeta, T* __restrict__ dv, int batch, int ch, int t, int pitch) {
  TS_ROW_UNIT((long long)batch * ch);
  const int c = row % ch;
  const float inv_n = 1.f / ((float)batch * (float)t);
  const float mg = dbeta[c] * inv_n, mgx = dgamma[c] * inv_n, mu = mean_rstd[2 * c], rs = mean_rstd[2 * c + 1], k = gamma[c] * rs;
  const size_t base = (size_t)row * pitch + i;
  float gg[8], vv[8];
  load8(g + base, gg);
  load8(v + base, vv);
#pragma unroll
  for (int j = 0; j < 8; ++j) gg[j] = k * (gg[j] - mg - (vv[j] - mu) * rs * mgx);
  store8(dv + base, gg);
}

// out = relu(a + b) (RELU) or a + b; backward of the first: da = db = dout * (out > 0)
template <class T, bool RELU>
__device__ __forceinline__ void add_rows(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long long rows, int t, int pitch,
                                         const int* __restrict__ len_b = nullptr, int ch = 1) {
  TS_ROW_UNIT(rows);
  const size_t base = (size_t)row * pitch + i;
  float x[8], z[8];
  load8(a + base, x);
  if (b) load8(b + base, z);
  if (len_b) {                                             // b counts only up to its clip's length (the mask of a MaskedConv1d input, backward)
    const int l = clamp_len(len_b, row / ch, t);
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = i + j < l ? z[j] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float s = x[j] + (b ? z[j] : 0.f); x[j] = (!RELU || s > 0.f) ? s : 0.f; }
  store8(o + base, x);
}
template <class T>
__global__ __launch_bounds__(256) void add_relu_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long long rows, int t, int pitch) {
  add_rows<T, true>(a, b, o, rows, t, pitch);
}
template <class T>
__global__ __launch_bounds__(256) void add_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long long rows, int t, int pitch,
                                                      const int* __restrict__ len_b, int ch) {
  add_rows<T, false>(a, b, o, rows, t, pitch, len_b, ch);
}
template <class T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ out, T* __restrict__ din, long long rows, int t, int pitch) {
  TS_ROW_UNIT(rows);
  const size_t base = (size_t)row * pitch + i;
  float g[8], o[8];
  load8(dout + base, g);
  load8(out + base, o);
#pragma unroll
  for (int j = 0; j < 8; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
  store8(din + base, g);
}

// sum of `parts` partial [rows] vectors (the per-clip dW of the pointwise backward)
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ parts, float* __restrict__ out, long long rows, int n_parts) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows) return;
  float s = 0.f;
  for (int i = 0; i < n_parts; ++i) s += parts[(size_t)i * rows + idx];
  out[idx] = s;
}

// reference-layout f32 [rows][t] (contiguous) <-> pitched activation rows of either type: the boundary of the training path
template <class T>
__global__ __launch_bounds__(256) void act_import_kernel(const float* __restrict__ src, T* __restrict__ dst, long long rows, int t, int pitch) {
  TS_ROW_UNIT(rows);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = i + j < t ? src[(size_t)row * t + i + j] : 0.f;
  store8(dst + (size_t)row * pitch + i, v);
}
template <class T>
__global__ __launch_bounds__(256) void act_export_kernel(const T* __restrict__ src, float* __restrict__ dst, long long rows, int t, int pitch) {
  TS_ROW_UNIT(rows);
  float v[8];
  load8(src + (size_t)row * pitch + i, v);
#pragma unroll
  for (int j = 0; j < 8; ++j) if (i + j < t) dst[(size_t)row * t + i + j] = v[j];
}

static inline unsigned blocks(long long n) { return (unsigned)((n + 255) / 256); }
static inline dim3 row_grid(long long rows, int t) { return dim3((unsigned)((rows * ((t + ROW_CHUNK - 1) / ROW_CHUNK) + 3) / 4)); }
static inline bool rows_ok(const void* p, int pitch, int act) {
  return pitch % 8 == 0 && reinterpret_cast<uintptr_t>(p) % (act ? 16 : 32) == 0;
}

}  // namespace ts

using namespace ts;
#define TS_STREAM hipStream_t stream = reinterpret_cast<hipStream_t>(stream_); (void)hipGetLastError()
// dispatch on the activation type: ACT(kernel, grid, block, lds, args...) launches kernel<float> or kernel<bf16_t>
// the "same" geometry the pair kernels cover
static bool pair_geometry(int ch, int t_in, int t_out, int k, int stride, int dil, int pad, int pitch_in, int pitch_out) {
  return stride == 1 && dil == 1 && (k & 1) && pad == (k - 1) / 2 && t_in == t_out && (ch & 1) == 0 && pitch_in == pitch_out && k <= DW_KMAX;
}

// dilation-2 "same" layers (QuartzNet's K87 block): the pair kernels in phase-split form
static bool phase_geometry(int t_in, int t_out, int k, int stride, int dil, int pad, int pitch_in, int pitch_out) {
  return stride == 1 && dil == 2 && (k & 1) && pad == k - 1 && (pad & 1) == 0 && t_in == t_out && pitch_in == pitch_out && k <= DW_KMAX;
}

#define TS_ACT(act, expr_f32, expr_bf16) do { if (act) { expr_bf16; } else { expr_f32; } } while (0)

static int dwconv_fwd_impl(const void* x, const int32_t* len_in, const int32_t* len_out, const float* w, void* y, int32_t batch,
                           int32_t ch, int32_t t_in, int32_t t_out, int32_t k, int32_t stride, int32_t dil, int32_t pad,
                           int32_t pitch_in, int32_t pitch_out, int32_t act, void* stream_, PairBnIn aff) {
  if (!x || !w || !y || batch <= 0 || ch <= 0 || t_in <= 0 || t_out <= 0 || k <= 0 || stride <= 0 || dil <= 0) return TS_EINVAL;
  if (pitch_in < t_in || pitch_out < t_out || act < 0 || act > 1) return TS_EINVAL;
  TS_STREAM;
  if (k > DW_KMAX) return TS_EUNSUPPORTED;
  if (aff.part && !pair_geometry(ch, t_in, t_out, k, stride, dil, pad, pitch_in, pitch_out)) return TS_EUNSUPPORTED;
  if (pair_geometry(ch, t_in, t_out, k, stride, dil, pad, pitch_in, pitch_out)) {
    if (act == 1 && batch >= 17) {
      // bf16 rows, enough clips to fill the MFMA's N dimension: the depthwise as Toeplitz x clips on the matrix cores
      const int pup = round_up(pad, 16), R = (k + 30 + pup - pad) / 16 + 1, NB = (R + 1) / 2;
      const int n_cg = (batch + 31) / 32, n_blk = (t_out + 31) / 32;
      int n_seg = (TS_DWM_WAVES_PER_CU * cu_count() + ch * n_cg - 1) / (ch * n_cg);   // ~2 waves per SIMD
      n_seg = n_seg < 1 ? 1 : (n_seg > n_blk ? n_blk : n_seg);
      const dim3 gridm((unsigned)(((long long)ch * n_cg * n_seg + 3) / 4));
#define TS_DWM(NB_) if (NB == NB_) { hipLaunchKernelGGL(dw_fwd_mfma_kernel<NB_>, gridm, dim3(256), 0, stream, (const bf16_t*)x, len_in, len_out, w, \
                                                        (bf16_t*)y, batch, ch, t_in, k, pad, pitch_in, n_seg, aff); return hip_status(hipGetLastError()); }
      TS_DWM(1) TS_DWM(2) TS_DWM(3) TS_DWM(4) TS_DWM(5)
#undef TS_DWM
    }
    if (aff.tiles) return TS_EUNSUPPORTED;                 // per-tile statistics: the matrix-core kernel only
    const size_t lds2 = 4 * (size_t)(pair_xl(round_up(k + 7, 8)) + PAIR_TAPS) * sizeof(v2f);
    // pairs per wave: 2 (the second pair's loads overlap the first one's FIR) once that still leaves >= 16 waves per CU
    const long long n_pairs = (long long)batch * ch / 2;
    const int ppw = n_pairs >= 32LL * cu_count() ? 2 : 1;
    const dim3 grid2((unsigned)((n_pairs + 4 * ppw - 1) / (4 * ppw)));
    TS_ACT(act,
           hipLaunchKernelGGL(dw_fwd_pair_kernel<float>, grid2, dim3(256), lds2, stream, (const float*)x, len_in, len_out, w, (float*)y, batch, ch,
                              t_in, k, pad, pitch_in, ppw, aff),
           hipLaunchKernelGGL(dw_fwd_pair_kernel<bf16_t>, grid2, dim3(256), lds2, stream, (const bf16_t*)x, len_in, len_out, w, (bf16_t*)y, batch,
                              ch, t_in, k, pad, pitch_in, ppw, aff));
    return hip_status(hipGetLastError());
  }
  if (phase_geometry(t_in, t_out, k, stride, dil, pad, pitch_in, pitch_out)) {
    const size_t lds2 = 4 * (size_t)(pair_xl(round_up(k + 7, 8)) + PAIR_TAPS) * sizeof(v2f);
    const long long n_rows = (long long)batch * ch;
    const int ppw = n_rows >= 32LL * cu_count() ? 2 : 1;
    const dim3 grid2((unsigned)((n_rows + 4 * ppw - 1) / (4 * ppw)));
    TS_ACT(act,
           { hipLaunchKernelGGL((dw_fwd_pair_kernel<float, true>), grid2, dim3(256), lds2, stream, (const float*)x, len_in, len_out, w, (float*)y, batch, ch,
                               t_in, k, pad / 2, pitch_in, ppw, aff); },
           { hipLaunchKernelGGL((dw_fwd_pair_kernel<bf16_t, true>), grid2, dim3(256), lds2, stream, (const bf16_t*)x, len_in, len_out, w, (bf16_t*)y, batch,
                               ch, t_in, k, pad / 2, pitch_in, ppw, aff); });
    return hip_status(hipGetLastError());
  }
  const size_t lds = (DW_KMAX + (size_t)(DW_TILE - 1) * stride + (size_t)(k - 1) * dil + 1 + 32) * sizeof(float);
  if (lds > 64 * 1024) return TS_EUNSUPPORTED;
  const dim3 grid((t_out + DW_TILE - 1) / DW_TILE, batch * ch);
  TS_ACT(act,
         hipLaunchKernelGGL(dw_fwd_kernel<float>, grid, dim3(256), lds, stream, (const float*)x, len_in, len_out, w, (float*)y, batch, ch, t_in,
                            t_out, k, stride, dil, pad, pitch_in, pitch_out),
         hipLaunchKernelGGL(dw_fwd_kernel<bf16_t>, grid, dim3(256), lds, stream, (const bf16_t*)x, len_in, len_out, w, (bf16_t*)y, batch, ch, t_in,
                            t_out, k, stride, dil, pad, pitch_in, pitch_out));
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_dwconv_fwd(const void* x, const int32_t* len_in, const int32_t* len_out, const float* w, void* y, int32_t batch,
                                   int32_t ch, int32_t t_in, int32_t t_out, int32_t k, int32_t stride, int32_t dil, int32_t pad,
                                   int32_t pitch_in, int32_t pitch_out, int32_t act, void* stream_) {
  return dwconv_fwd_impl(x, len_in, len_out, w, y, batch, ch, t_in, t_out, k, stride, dil, pad, pitch_in, pitch_out, act, stream_,
                         PairBnIn{nullptr, nullptr, nullptr, 0.f, 0, 1.0, nullptr, nullptr, nullptr, 0.f, nullptr});
}

// the same with x = relu?(BatchNorm(v)) formed on the fly from the previous repeat's un-normalised output v and the clip-group sums
// ts_train_bn_stats left in `in_sums`; publishes in_mean_rstd (f32 [C][2], for the backward pass) and updates the running statistics
extern "C" int ts_train_dwconv_fwd_bn_tiles(const void* v, const float* in_tile_sums, int32_t in_tiles, const float* in_gamma, const float* in_beta, float in_eps,
                                            int32_t in_relu, float* in_mean_rstd, float* running_mean, float* running_var, float momentum,
                                            int64_t* num_batches_tracked, const int32_t* len_in, const int32_t* len_out, const float* w, void* y,
                                            int32_t batch, int32_t ch, int32_t t, int32_t k, int32_t pad, int32_t pitch, int32_t act, void* stream_) {
  if (!in_tile_sums || in_tiles <= 0 || !in_gamma || !in_beta || !in_mean_rstd || (running_mean == nullptr) != (running_var == nullptr)) return TS_EINVAL;
  // only the matrix-core kernel sums tile pairs (a wave per channel); dwconv_fwd_impl takes it for bf16 rows, >= 17 clips, "same" geometry
  if (act != 1 || batch < 17 || !pair_geometry(ch, t, t, k, 1, 1, pad, pitch, pitch)) return TS_EUNSUPPORTED;
  PairBnIn in{nullptr, in_gamma, in_beta, in_eps, in_relu, (double)batch * t, in_mean_rstd, running_mean, running_var, momentum,
              reinterpret_cast<long long*>(num_batches_tracked)};
  in.tiles = in_tile_sums; in.n_tiles = in_tiles;
  return dwconv_fwd_impl(v, len_in, len_out, w, y, batch, ch, t, t, k, 1, 1, pad, pitch, pitch, act, stream_, in);
}

extern "C" int ts_train_dwconv_fwd_bn(const void* v, const void* in_sums, const float* in_gamma, const float* in_beta, float in_eps,
                                      int32_t in_relu, float* in_mean_rstd, float* running_mean, float* running_var, float momentum,
                                      int64_t* num_batches_tracked, const int32_t* len_in, const int32_t* len_out, const float* w, void* y,
                                      int32_t batch, int32_t ch, int32_t t, int32_t k, int32_t pad, int32_t pitch, int32_t act, void* stream_) {
  if (!in_sums || !in_gamma || !in_beta || !in_mean_rstd || (running_mean == nullptr) != (running_var == nullptr)) return TS_EINVAL;
  return dwconv_fwd_impl(v, len_in, len_out, w, y, batch, ch, t, t, k, 1, 1, pad, pitch, pitch, act, stream_,
                         PairBnIn{static_cast<const double*>(in_sums), in_gamma, in_beta, in_eps, in_relu, (double)batch * t, in_mean_rstd,
                                  running_mean, running_var, momentum, reinterpret_cast<long long*>(num_batches_tracked)});
}

// the matrix-core depthwise backward (dw_bwd_mfma_kernel): bf16 rows, "same" geometry, channels % 16 == 0, odd K <= 75
static int g_dw_bwd_mfma = 1, g_dw_bwd_tile = 128;
/* 1 (default): bf16 depthwise backward on the matrix cores where the geometry allows; 0: always the VALU pair kernel (A/B, tests); 2: as 1 on
   256-frame tiles (one workgroup per compute unit) */
extern "C" int ts_train_dwconv_bwd_select(int32_t mode) {
  const int old = g_dw_bwd_mfma ? (g_dw_bwd_tile == 256 ? 2 : 1) : 0;
  g_dw_bwd_mfma = mode ? 1 : 0;
  g_dw_bwd_tile = mode == 2 ? 256 : 128;
  return old;
}

// Deterministic mode (ts_train_set_deterministic): the depthwise backward kernels' per-workgroup sums go to a caller-provided workspace
// [n_parts][ch][k + 2] (taps, then dbeta, dgamma) instead of float atomics, and this kernel adds them to their destinations in workgroup order --
// the step's gradients are then a pure function of its inputs (same bits on every run and box), at one extra launch per layer.
static float* g_det_ws = nullptr;
static long long g_det_floats = 0;
__global__ __launch_bounds__(256) void det_reduce_kernel(const float* __restrict__ part, int n_parts, int ch, int k, float* __restrict__ dw,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= ch * (k + 2)) return;
  const int c = idx / (k + 2), j = idx % (k + 2);
  float* dst = j < k ? (dw ? dw + (size_t)c * k + j : nullptr) : (j == k ? (dbeta ? dbeta + c : nullptr) : (dgamma ? dgamma + c : nullptr));
  if (!dst) return;
  float tot = 0.f;
  for (int y = 0; y < n_parts; ++y) tot += part[((size_t)y * ch + c) * (k + 2) + j];
  *dst += tot;
}
// workspace for a launch of n_parts x ch x (k + 2) partials, or null (not in deterministic mode); *st = TS_EINVAL when the workspace is too small
static float* det_workspace(int n_parts, int ch, int k, int* st) {
  if (!g_det_ws) return nullptr;
  if ((long long)n_parts * ch * (k + 2) > g_det_floats) { *st = TS_EINVAL; return nullptr; }
  return g_det_ws;
}
static int det_reduce(const float* part, int n_parts, int ch, int k, float* dw, float* dgamma, float* dbeta, hipStream_t stream) {
  hipLaunchKernelGGL(det_reduce_kernel, dim3((ch * (k + 2) + 255) / 256), dim3(256), 0, stream, part, n_parts, ch, k, dw, dgamma, dbeta);
  return hip_status(hipGetLastError());
}
extern "C" int ts_train_set_deterministic(float* workspace, int64_t n_floats) {
  if (workspace && n_floats <= 0) return TS_EINVAL;
  g_det_ws = workspace;
  g_det_floats = workspace ? n_floats : 0;
  return TS_OK;
}

template <int TT, int NK, int G>
static int dw_bwd_mfma_go(DwbArgs& a, int n_cg, hipStream_t stream) {
  constexpr int WY = TT + 96, WX = TT + 96;
  const size_t lds = (size_t)4 * (16 * dwb_pitch(WY, 4) * 2 + 16 * dwb_pitch(WX, 8) * 2);
  auto kern = dw_bwd_mfma_kernel<TT, NK, G>;
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TS_EINVAL;
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return TS_EUNSUPPORTED;
    attr_set[dev] = true;
  }
  const int units = a.batch * a.n_tiles, ny = (units + 4 * a.upw - 1) / (4 * a.upw);
  int st = TS_OK;
  a.part = det_workspace(ny, a.ch, a.k, &st);
  if (st != TS_OK) return st;
  hipLaunchKernelGGL(kern, dim3(n_cg, ny), dim3(256), lds, stream, a);
  st = hip_status(hipGetLastError());
  if (st == TS_OK && a.part) st = det_reduce(a.part, ny, a.ch, a.k, a.dw, a.aff.mean_rstd ? a.in_dgamma : nullptr, a.aff.mean_rstd ? a.in_dbeta : nullptr, stream);
  return st;
}

static int dw_bwd_mfma_launch(const void* dy, const void* x, const int32_t* len_in, const int32_t* len_out, const float* w, void* dx, float* dw,
                              int batch, int ch, int t, int k, int pad, int pitch, PairAffine aff, float* in_dgamma, float* in_dbeta, hipStream_t stream) {
  if (ch % 16 || !(k & 1) || k > 75 || k < 3) return TS_EUNSUPPORTED;
  const int p4 = round_up(pad, 4), dl = p4 - pad;
  const int nk = (dl + k + 2) / 4 + 1, g = (dl + k + 15) / 16;
  DwbArgs a;
  a.dy = (const bf16_t*)dy; a.x = (const bf16_t*)x; a.len_in = len_in; a.len_out = len_out; a.w = w; a.dx = (bf16_t*)dx; a.dw = dw;
  a.batch = batch; a.ch = ch; a.t = t; a.k = k; a.p = pad; a.pitch = pitch; a.aff = aff; a.in_dgamma = in_dgamma; a.in_dbeta = in_dbeta;
  // 128-frame tiles: 62 KiB of LDS per workgroup, two workgroups (8 waves) per compute unit
  const int TT = g_dw_bwd_tile;
  a.n_tiles = (t + TT - 1) / TT;
  // units per wave: enough workgroups for every compute unit, as few atomics per (channel, tap) as that allows
  const int n_cg = ch / 16, units = batch * a.n_tiles, wg_per_cu = TT <= 128 ? 2 : 1;
  int upw = (int)(((long long)n_cg * units + 4LL * wg_per_cu * cu_count() - 1) / (4LL * wg_per_cu * cu_count()));
  a.upw = upw < 1 ? 1 : upw;
  (void)hipGetLastError();
#define TS_DWB(NK_, G_) if (nk <= NK_ && g <= G_) return TT == 128 ? dw_bwd_mfma_go<128, NK_, G_>(a, n_cg, stream) : dw_bwd_mfma_go<256, NK_, G_>(a, n_cg, stream);
  TS_DWB(9, 3) TS_DWB(11, 3) TS_DWB(15, 4) TS_DWB(17, 4) TS_DWB(21, 5)
#undef TS_DWB
  return TS_EUNSUPPORTED;
}

static int dwconv_bwd_impl(const void* dy, const void* x, const int32_t* len_in, const int32_t* len_out, const float* w,
                           void* dx, float* dw, int32_t batch, int32_t ch, int32_t t_in, int32_t t_out, int32_t k,
                           int32_t stride, int32_t dil, int32_t pad, int32_t pitch_in, int32_t pitch_out, int32_t act, void* stream_,
                           PairAffine aff, float* in_dgamma, float* in_dbeta) {
  if (!dy || !x || !w || batch <= 0 || ch <= 0 || t_in <= 0 || t_out <= 0 || k <= 0) return TS_EINVAL;      // dw may be null: frozen weight
  // dx may be null where the input needs no gradient (the stem's input are the features) -- the general-geometry kernels only, whose data and
  // weight gradients are separate launches; the fused kernels form both in one pass and want the buffer
  const bool fused = pair_geometry(ch, t_in, t_out, k, stride, dil, pad, pitch_in, pitch_out) || phase_geometry(t_in, t_out, k, stride, dil, pad, pitch_in, pitch_out);
  if (!dx && (fused || !dw)) return TS_EINVAL;
  if (pitch_in < t_in || pitch_out < t_out || act < 0 || act > 1) return TS_EINVAL;
  TS_STREAM;
  if (k > DW_KMAX) return TS_EUNSUPPORTED;
  if (aff.mean_rstd && !pair_geometry(ch, t_in, t_out, k, stride, dil, pad, pitch_in, pitch_out)) return TS_EUNSUPPORTED;
  if (pair_geometry(ch, t_in, t_out, k, stride, dil, pad, pitch_in, pitch_out)) {
    if (act == 1 && g_dw_bwd_mfma) {
      const int st = dw_bwd_mfma_launch(dy, x, len_in, len_out, w, dx, dw, batch, ch, t_in, k, pad, pitch_in, aff, in_dgamma, in_dbeta, stream);
      if (st != TS_EUNSUPPORTED) return st;
    }
    const size_t lds2 = 4 * (size_t)(pair_xl(round_up(k + 7, 8)) + pair_gl(k, pad) + PAIR_TAPS) * sizeof(v2f);
    const int cpw = batch >= 16 ? (batch + 15) / 16 : 1;        // >= 2 clips per wave: the second clip's loads overlap the first one's FIR
    const dim3 grid2(ch / 2, (batch + 4 * cpw - 1) / (4 * cpw));
    int dst = TS_OK;
    float* const part = det_workspace((int)grid2.y, ch, k, &dst);
    if (dst != TS_OK) return dst;
    TS_ACT(act,
           hipLaunchKernelGGL(dw_bwd_pair_kernel<float>, grid2, dim3(256), lds2, stream, (const float*)dy, (const float*)x, len_in, len_out, w,
                              (float*)dx, dw, batch, ch, t_in, k, pad, pitch_in, cpw, aff, in_dgamma, in_dbeta, part),
           hipLaunchKernelGGL(dw_bwd_pair_kernel<bf16_t>, grid2, dim3(256), lds2, stream, (const bf16_t*)dy, (const bf16_t*)x, len_in, len_out, w,
                              (bf16_t*)dx, dw, batch, ch, t_in, k, pad, pitch_in, cpw, aff, in_dgamma, in_dbeta, part));
    dst = hip_status(hipGetLastError());
    if (dst == TS_OK && part) dst = det_reduce(part, (int)grid2.y, ch, k, dw, aff.mean_rstd ? in_dgamma : nullptr, aff.mean_rstd ? in_dbeta : nullptr, stream);
    return dst;
  }
  if (phase_geometry(t_in, t_out, k, stride, dil, pad, pitch_in, pitch_out)) {
    const size_t lds2 = 4 * (size_t)(pair_xl(round_up(k + 7, 8)) + pair_gl(k, pad / 2) + PAIR_TAPS) * sizeof(v2f);
    const int cpw = batch >= 16 ? (batch + 15) / 16 : 1;
    const dim3 grid2(ch, (batch + 4 * cpw - 1) / (4 * cpw));
    int dst = TS_OK;
    float* const part = det_workspace((int)grid2.y, ch, k, &dst);
    if (dst != TS_OK) return dst;
    TS_ACT(act,
           { hipLaunchKernelGGL((dw_bwd_pair_kernel<float, true>), grid2, dim3(256), lds2, stream, (const float*)dy, (const float*)x, len_in, len_out, w,
                               (float*)dx, dw, batch, ch, t_in, k, pad / 2, pitch_in, cpw, aff, in_dgamma, in_dbeta, part); },
           { hipLaunchKernelGGL((dw_bwd_pair_kernel<bf16_t, true>), grid2, dim3(256), lds2, stream, (const bf16_t*)dy, (const bf16_t*)x, len_in, len_out, w,
                               (bf16_t*)dx, dw, batch, ch, t_in, k, pad / 2, pitch_in, cpw, aff, in_dgamma, in_dbeta, part); });
    dst = hip_status(hipGetLastError());
    if (dst == TS_OK && part) dst = det_reduce(part, (int)grid2.y, ch, k, dw, nullptr, nullptr, stream);
    return dst;
  }
  const size_t lds_d = (DW_KMAX + (size_t)(DW_TILE + (k - 1) * dil) / stride + 2 + 48) * sizeof(float);
  const size_t lds_w = (size_t)(round_up(t_out + 16, 4) + round_up(t_in + 2 * pad + 48, 2)) * sizeof(float) + 256 * 8 * sizeof(double);
  if (k > 256) return TS_EUNSUPPORTED;
  if (lds_d > 64 * 1024 || lds_w > 64 * 1024) return TS_EUNSUPPORTED;
  const dim3 gd((t_in + DW_TILE - 1) / DW_TILE, batch * ch), gw(ch, batch < 8 ? batch : 8);
  if (dx)
  TS_ACT(act,
         hipLaunchKernelGGL(dw_bwd_data_kernel<float>, gd, dim3(256), lds_d, stream, (const float*)dy, len_in, len_out, w, (float*)dx, batch, ch,
                            t_in, t_out, k, stride, dil, pad, pitch_in, pitch_out),
         hipLaunchKernelGGL(dw_bwd_data_kernel<bf16_t>, gd, dim3(256), lds_d, stream, (const bf16_t*)dy, len_in, len_out, w, (bf16_t*)dx, batch, ch,
                            t_in, t_out, k, stride, dil, pad, pitch_in, pitch_out));
  if (dw) {
    int dst = TS_OK;
    float* const part = det_workspace((int)gw.y, ch, k, &dst);
    if (dst != TS_OK) return dst;
    TS_ACT(act,
           hipLaunchKernelGGL(dw_bwd_weight_kernel<float>, gw, dim3(256), lds_w, stream, (const float*)dy, (const float*)x, len_in, len_out, dw, batch,
                              ch, t_in, t_out, k, stride, dil, pad, pitch_in, pitch_out, part),
           hipLaunchKernelGGL(dw_bwd_weight_kernel<bf16_t>, gw, dim3(256), lds_w, stream, (const bf16_t*)dy, (const bf16_t*)x, len_in, len_out, dw, batch,
                              ch, t_in, t_out, k, stride, dil, pad, pitch_in, pitch_out, part));
    if (part) {
      dst = hip_status(hipGetLastError());
      if (dst != TS_OK) return dst;
      return det_reduce(part, (int)gw.y, ch, k, dw, nullptr, nullptr, stream);
    }
  }
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_dwconv_bwd(const void* dy, const void* x, const int32_t* len_in, const int32_t* len_out, const float* w,
                                   void* dx, float* dw, int32_t batch, int32_t ch, int32_t t_in, int32_t t_out, int32_t k,
                                   int32_t stride, int32_t dil, int32_t pad, int32_t pitch_in, int32_t pitch_out, int32_t act, void* stream_) {
  return dwconv_bwd_impl(dy, x, len_in, len_out, w, dx, dw, batch, ch, t_in, t_out, k, stride, dil, pad, pitch_in, pitch_out, act, stream_,
                         PairAffine{nullptr, nullptr, nullptr, 0}, nullptr, nullptr);
}

// backward of ts_train_dwconv_fwd_bn: dw as above; `g` receives dL/dy * (y > 0) (y = the transformed input) and in_dbeta / in_dgamma
// ACCUMULATE sum g and sum g * xhat per channel: the first half of the previous BatchNorm's backward (ts_train_bn_bwd_sums is the second)
extern "C" int ts_train_dwconv_bwd_bn(const void* dy, const void* v, const float* in_mean_rstd, const float* in_gamma, const float* in_beta,
                                      int32_t in_relu, const int32_t* len_in, const int32_t* len_out, const float* w, void* g, float* dw,
                                      float* in_dgamma, float* in_dbeta, int32_t batch, int32_t ch, int32_t t, int32_t k, int32_t pad,
                                      int32_t pitch, int32_t act, void* stream_) {
  if (!in_mean_rstd || !in_gamma || !in_beta || !in_dgamma || !in_dbeta) return TS_EINVAL;
  return dwconv_bwd_impl(dy, v, len_in, len_out, w, g, dw, batch, ch, t, t, k, 1, 1, pad, pitch, pitch, act, stream_,
                         PairAffine{in_mean_rstd, in_gamma, in_beta, in_relu}, in_dgamma, in_dbeta);
}

// BatchNorm(train) batch sums without the apply pass: sums = double [8 clip groups][C][2] (sum v, sum v^2), consumed by
// ts_train_dwconv_fwd_bn
extern "C" int ts_train_bn_stats(const void* v, void* sums, int32_t batch, int32_t ch, int32_t t, int32_t pitch, int32_t act, void* stream_) {
  if (!v || !sums || batch <= 0 || ch <= 0 || t <= 0 || act < 0 || act > 1 || !rows_ok(v, pitch, act) || pitch < t) return TS_EINVAL;
  TS_STREAM;
  double* part = static_cast<double*>(sums);
  TS_ACT(act,
         hipLaunchKernelGGL((chan_sums_kernel<0, float>), dim3(ch, BN_G), dim3(256), 0, stream, (const float*)v, (const float*)nullptr, (const float*)nullptr,
                            (const float*)nullptr, part, batch, ch, t, pitch, 0),
         hipLaunchKernelGGL((chan_sums_kernel<0, bf16_t>), dim3(ch, BN_G), dim3(256), 0, stream, (const bf16_t*)v, (const bf16_t*)nullptr, (const bf16_t*)nullptr,
                            (const float*)nullptr, part, batch, ch, t, pitch, 0));
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_bn2_add_relu_fwd(const void* va, const void* sums_a, const float* gamma_a, const float* beta_a, float eps_a,
                                         float* mean_rstd_a, float* running_mean_a, float* running_var_a, float momentum_a, int64_t* nbt_a,
                                         const void* vb, const void* sums_b, const float* gamma_b, const float* beta_b, float eps_b,
                                         float* mean_rstd_b, float* running_mean_b, float* running_var_b, float momentum_b, int64_t* nbt_b,
                                         void* out, int32_t batch, int32_t ch, int32_t t, int32_t pitch, int32_t act, void* stream_) {
  if (!va || !sums_a || !gamma_a || !beta_a || !mean_rstd_a || !vb || !sums_b || !gamma_b || !beta_b || !mean_rstd_b || !out) return TS_EINVAL;
  if (batch <= 0 || ch <= 0 || t <= 0 || act < 0 || act > 1 || pitch < t) return TS_EINVAL;
  if ((running_mean_a == nullptr) != (running_var_a == nullptr) || (running_mean_b == nullptr) != (running_var_b == nullptr)) return TS_EINVAL;
  if (!rows_ok(va, pitch, act) || !rows_ok(vb, pitch, act) || !rows_ok(out, pitch, act)) return TS_EINVAL;
  TS_STREAM;
  const BnSide a{va, static_cast<const double*>(sums_a), gamma_a, beta_a, eps_a, mean_rstd_a, running_mean_a, running_var_a, momentum_a,
                 reinterpret_cast<long long*>(nbt_a)};
  const BnSide b{vb, static_cast<const double*>(sums_b), gamma_b, beta_b, eps_b, mean_rstd_b, running_mean_b, running_var_b, momentum_b,
                 reinterpret_cast<long long*>(nbt_b)};
  const dim3 rg = row_grid((long long)batch * ch, t);
  TS_ACT(act,
         hipLaunchKernelGGL(bn2_add_relu_kernel<float>, rg, dim3(256), 0, stream, a, b, (float*)out, batch, ch, t, pitch),
         hipLaunchKernelGGL(bn2_add_relu_kernel<bf16_t>, rg, dim3(256), 0, stream, a, b, (bf16_t*)out, batch, ch, t, pitch));
  return hip_status(hipGetLastError());
}

/* Block tail forward without separate statistics passes (one workgroup per channel, rows in registers); TS_EUNSUPPORTED when the batch does not
 * fit the register budget (callers then run ts_train_bn_stats x 2 + ts_train_bn2_add_relu_fwd).  See include/thunder_speech_amd.h */
extern "C" int ts_train_bn2_add_relu_chan_fwd(const void* va, const float* gamma_a, const float* beta_a, float eps_a, float* mean_rstd_a,
                                              float* running_mean_a, float* running_var_a, float momentum_a, int64_t* nbt_a, const void* vb,
                                              const float* gamma_b, const float* beta_b, float eps_b, float* mean_rstd_b, float* running_mean_b,
                                              float* running_var_b, float momentum_b, int64_t* nbt_b, void* out, int32_t batch, int32_t ch, int32_t t,
                                              int32_t pitch, int32_t act, void* stream_) {
  if (!va || !vb || !gamma_a || !beta_a || !gamma_b || !beta_b || !mean_rstd_a || !mean_rstd_b || !out) return TS_EINVAL;
  if (batch <= 0 || ch <= 0 || t <= 0 || pitch < t || pitch % 8 || act < 0 || act > 1) return TS_EINVAL;
  const int units = batch * ((t + ROW_CHUNK - 1) / ROW_CHUNK);
  if (units > 4 * (act ? ChanRegs<bf16_t>::UMAX : ChanRegs<float>::UMAX) || (long long)batch * ch * pitch >= (1ll << 31)) return TS_EUNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  Bn2FwdArgs g{};
  g.a = BnSide{va, nullptr, gamma_a, beta_a, eps_a, mean_rstd_a, running_mean_a, running_var_a, momentum_a, (long long*)nbt_a};
  g.b = BnSide{vb, nullptr, gamma_b, beta_b, eps_b, mean_rstd_b, running_mean_b, running_var_b, momentum_b, (long long*)nbt_b};
  g.out = out; g.batch = batch; g.ch = ch; g.t = t; g.pitch = pitch;
  (void)hipGetLastError();
  TS_ACT(act, hipLaunchKernelGGL(bn2_fwd_chan_kernel<float>, dim3(ch), dim3(256), 0, stream, g),
         hipLaunchKernelGGL(bn2_fwd_chan_kernel<bf16_t>, dim3(ch), dim3(256), 0, stream, g));
  return hip_status(hipGetLastError());
}

/* Backward of the block tail, both branches, one launch (same budget rule) */
extern "C" int ts_train_bn2_chan_bwd(const void* dout, const void* dout2, const int32_t* len2, const void* out, const void* va, const void* vb, const float* gamma_a, const float* mean_rstd_a,
                                     const float* gamma_b, const float* mean_rstd_b, void* dva, void* dvb, float* dgamma_a, float* dbeta_a,
                                     float* dgamma_b, float* dbeta_b, int32_t batch, int32_t ch, int32_t t, int32_t pitch, int32_t act, void* stream_) {
  if (!dout || !out || !va || !vb || !gamma_a || !gamma_b || !mean_rstd_a || !mean_rstd_b || !dva || !dvb || !dgamma_a || !dbeta_a || !dgamma_b || !dbeta_b)
    return TS_EINVAL;
  if (batch <= 0 || ch <= 0 || t <= 0 || pitch < t || pitch % 8 || act < 0 || act > 1) return TS_EINVAL;
  const int units = batch * ((t + ROW_CHUNK - 1) / ROW_CHUNK);
  if (units > 4 * (act ? ChanRegs<bf16_t>::UMAX : ChanRegs<float>::UMAX) || (long long)batch * ch * pitch >= (1ll << 31)) return TS_EUNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  const Bn2BwdArgs g{dout, dout2, len2, out, va, vb, gamma_a, mean_rstd_a, gamma_b, mean_rstd_b, dva, dvb, dgamma_a, dbeta_a, dgamma_b, dbeta_b, batch, ch, t, pitch};
  (void)hipGetLastError();
  TS_ACT(act, hipLaunchKernelGGL(bn2_bwd_chan_kernel<float>, dim3(ch), dim3(256), 0, stream, g),
         hipLaunchKernelGGL(bn2_bwd_chan_kernel<bf16_t>, dim3(ch), dim3(256), 0, stream, g));
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_bn_bwd_sums(const void* g, const void* v, const float* gamma, const float* mean_rstd, const float* dgamma,
                                    const float* dbeta, void* dv, int32_t batch, int32_t ch, int32_t t, int32_t pitch, int32_t act, void* stream_) {
  if (!g || !v || !gamma || !mean_rstd || !dgamma || !dbeta || !dv || batch <= 0 || ch <= 0 || t <= 0 || act < 0 || act > 1) return TS_EINVAL;
  if (!rows_ok(g, pitch, act) || !rows_ok(v, pitch, act) || !rows_ok(dv, pitch, act) || pitch < t) return TS_EINVAL;
  TS_STREAM;
  const dim3 rg = row_grid((long long)batch * ch, t);
  TS_ACT(act,
         hipLaunchKernelGGL(bn_bwd_sums_kernel<float>, rg, dim3(256), 0, stream, (const float*)g, (const float*)v, gamma, mean_rstd, dgamma, dbeta, (float*)dv,
                            batch, ch, t, pitch),
         hipLaunchKernelGGL(bn_bwd_sums_kernel<bf16_t>, rg, dim3(256), 0, stream, (const bf16_t*)g, (const bf16_t*)v, gamma, mean_rstd, dgamma, dbeta, (bf16_t*)dv,
                            batch, ch, t, pitch));
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_mask_time(const void* x, const int32_t* len, void* y, int32_t batch, int32_t ch, int32_t t, int32_t pitch_x,
                                  int32_t pitch_y, int32_t act, void* stream_) {
  if (!x || !len || !y || batch <= 0 || ch <= 0 || t <= 0 || act < 0 || act > 1) return TS_EINVAL;
  if (!rows_ok(x, pitch_x, act) || !rows_ok(y, pitch_y, act) || pitch_x < t || pitch_y < t) return TS_EINVAL;
  TS_STREAM;
  TS_ACT(act,
         hipLaunchKernelGGL(mask_time_kernel<float>, row_grid((long long)batch * ch, t), dim3(256), 0, stream, (const float*)x, len, (float*)y, batch, ch, t, pitch_x, pitch_y),
         hipLaunchKernelGGL(mask_time_kernel<bf16_t>, row_grid((long long)batch * ch, t), dim3(256), 0, stream, (const bf16_t*)x, len, (bf16_t*)y, batch, ch, t, pitch_x, pitch_y));
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_act_import(const float* src, void* dst, int64_t rows, int32_t t, int32_t pitch, int32_t act, void* stream_) {
  if (!src || !dst || rows <= 0 || t <= 0 || pitch < t || act < 0 || act > 1 || !rows_ok(dst, pitch, act)) return TS_EINVAL;
  TS_STREAM;
  TS_ACT(act,
         hipLaunchKernelGGL(act_import_kernel<float>, row_grid(rows, t), dim3(256), 0, stream, src, (float*)dst, (long long)rows, t, pitch),
         hipLaunchKernelGGL(act_import_kernel<bf16_t>, row_grid(rows, t), dim3(256), 0, stream, src, (bf16_t*)dst, (long long)rows, t, pitch));
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_act_export(const void* src, float* dst, int64_t rows, int32_t t, int32_t pitch, int32_t act, void* stream_) {
  if (!src || !dst || rows <= 0 || t <= 0 || pitch < t || act < 0 || act > 1 || !rows_ok(src, pitch, act)) return TS_EINVAL;
  TS_STREAM;
  TS_ACT(act,
         hipLaunchKernelGGL(act_export_kernel<float>, row_grid(rows, t), dim3(256), 0, stream, (const float*)src, dst, (long long)rows, t, pitch),
         hipLaunchKernelGGL(act_export_kernel<bf16_t>, row_grid(rows, t), dim3(256), 0, stream, (const bf16_t*)src, dst, (long long)rows, t, pitch));
  return hip_status(hipGetLastError());
}

// fp32 -> bf16 (round to nearest even): operand copies of the weights for the bf16 GEMMs
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long long n) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    *reinterpret_cast<uint2*>(y + i) = uint2{pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)};
  } else {
    for (long long j = i; j < n; ++j) y[j] = (unsigned short)(pack_bf16(x[j], 0.f) & 0xffffu);
  }
}

namespace ts {
// f32-accumulating GEMM on the f32 matrix-core instruction, any operand layout (csrc/gemm_f32.hip)
int gemm_f32(hipStream_t stream, bool in_bf16, const void* a, long long a_rs, long long a_cs, long long sa, long long ska, const void* b,
             long long b_rs, long long b_cs, long long sb, long long skb, void* c, long long ldc, long long sc, bool out_bf16, const float* bias,
             int M, int N, int K, int nkb, int batch, bool beta);
}

extern "C" int ts_train_cast_bf16(const float* x, void* y, int64_t n, void* stream_) {
  if (!x || !y || n <= 0) return TS_EINVAL;
  if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(y) % 8) return TS_EINVAL;
  TS_STREAM;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks((n + 3) / 4)), dim3(256), 0, stream, x, static_cast<unsigned short*>(y), (long long)n);
  return hip_status(hipGetLastError());
}

// v[b] = W . u[b]   (W [c_out][c_in] row-major, u [B][c_in][pitch_u], v [B][c_out][pitch_v]); u is expected masked by the caller.
// precision 0: f32 operands and result; 1: u and w bf16, v f32 (the decoder's logits); 2: u, w and v bf16.  f32 accumulation always.
extern "C" int ts_train_pwconv_fwd(const void* u, const void* w, void* v, int32_t batch, int32_t c_in, int32_t c_out, int32_t t,
                                   int32_t pitch_u, int32_t pitch_v, int32_t precision, void* stream_) {
  if (!u || !w || !v || batch <= 0 || c_in <= 0 || c_out <= 0 || t <= 0 || pitch_u < t || pitch_v < t) return TS_EINVAL;
  if (precision < 0 || precision > 2) return TS_EUNSUPPORTED;
  TS_STREAM;
  // per clip: V[c_out][t] = W[c_out][c_in] . U[c_in][t]  (W's contraction index contiguous, U's frame index contiguous)
  return gemm_f32(stream, precision != 0, w, c_in, 1, 0, 0, u, pitch_u, 1, (long long)c_in * pitch_u, 0, v, pitch_v, (long long)c_out * pitch_v,
                  precision == 2, nullptr, c_out, t, c_in, 1, batch, false);
}

// du[b] = W^T . dv[b];  dW = sum_b dv[b] . u[b]^T  (workspace: batch * c_out * c_in floats, f32 always); precision as above
// (1: dv, u, w bf16 and du f32; 2: du bf16 as well)
extern "C" int ts_train_pwconv_bwd(const void* dv, const void* u, const void* w, void* du, float* dw, float* workspace, int32_t batch,
                                   int32_t c_in, int32_t c_out, int32_t t, int32_t pitch_u, int32_t pitch_v, int32_t precision, void* stream_) {
  if (!dv || !u || !w || !du || !dw || !workspace || batch <= 0 || c_in <= 0 || c_out <= 0 || t <= 0 || pitch_u < t || pitch_v < t) return TS_EINVAL;
  if (precision < 0 || precision > 2) return TS_EUNSUPPORTED;
  TS_STREAM;
  const bool bf = precision != 0;
  // per clip: dU[c_in][t] = W^T . dV[c_out][t]  (A(m, k) = W[k][m]: W's output index is the contiguous one here)
  if (int st = gemm_f32(stream, bf, w, 1, c_in, 0, 0, dv, pitch_v, 1, (long long)c_out * pitch_v, 0, du, pitch_u, (long long)c_in * pitch_u,
                        precision == 2, nullptr, c_in, t, c_out, 1, batch, false))
    return st;
  // per clip: dW_b[c_out][c_in] = dV[c_out][t] . U[c_in][t]^T (both contract over their contiguous frame index) -> workspace, summed below.
  // One partial per clip keeps every CU busy (16 tiles x 32 clips at 512 x 512); the pitch padding beyond t never enters (K = t).
  if (int st = gemm_f32(stream, bf, dv, pitch_v, 1, (long long)c_out * pitch_v, 0, u, 1, pitch_u, (long long)c_in * pitch_u, 0, workspace, c_in,
                        (long long)c_in * c_out, false, nullptr, c_out, c_in, t, 1, batch, false))
    return st;
  const long long rows = (long long)c_in * c_out;
  hipLaunchKernelGGL(sum_parts_kernel, dim3(blocks(rows)), dim3(256), 0, stream, workspace, dw, rows, batch);
  return hip_status(hipGetLastError());
}

// workspace: 16 * c doubles (8 clip-group partials of 2 sums).  mean_rstd f32 [c][2] is saved for the backward.
extern "C" int ts_train_bn_fwd(const void* v, const float* gamma, const float* beta, void* y, float* mean_rstd, void* workspace,
                               int32_t batch, int32_t ch, int32_t t, int32_t pitch, float eps, int32_t relu, float* running_mean,
                               float* running_var, float momentum, int64_t* num_batches_tracked, int32_t act, void* stream_) {
  if (!v || !gamma || !beta || !y || !mean_rstd || !workspace || batch <= 0 || ch <= 0 || t <= 0 || act < 0 || act > 1) return TS_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return TS_EINVAL;
  if (!rows_ok(v, pitch, act) || !rows_ok(y, pitch, act) || pitch < t) return TS_EINVAL;
  TS_STREAM;
  double* sums = static_cast<double*>(workspace);
  long long* nbt = reinterpret_cast<long long*>(num_batches_tracked);
  const dim3 rg = row_grid((long long)batch * ch, t);
  const int ng = BN_G;
  TS_ACT(act,
         { hipLaunchKernelGGL((chan_sums_kernel<0, float>), dim3(ch, BN_G), dim3(256), 0, stream, (const float*)v, (const float*)nullptr, (const float*)nullptr,
                                (const float*)nullptr, sums, batch, ch, t, pitch, 0);
           hipLaunchKernelGGL(bn_fwd_kernel<float>, rg, dim3(256), 0, stream, (const float*)v, sums, gamma, beta, (float*)y, mean_rstd, batch, ch, t, pitch,
                              eps, relu, running_mean, running_var, momentum, nbt, ng); },
         { hipLaunchKernelGGL((chan_sums_kernel<0, bf16_t>), dim3(ch, BN_G), dim3(256), 0, stream, (const bf16_t*)v, (const bf16_t*)nullptr, (const bf16_t*)nullptr,
                                (const float*)nullptr, sums, batch, ch, t, pitch, 0);
           hipLaunchKernelGGL(bn_fwd_kernel<bf16_t>, rg, dim3(256), 0, stream, (const bf16_t*)v, sums, gamma, beta, (bf16_t*)y, mean_rstd, batch, ch, t, pitch,
                              eps, relu, running_mean, running_var, momentum, nbt, ng); });
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_bn_bwd(const void* dy, const void* y, const void* v, const float* gamma, const float* mean_rstd, void* dv,
                               float* dgamma, float* dbeta, void* workspace, int32_t batch, int32_t ch, int32_t t, int32_t pitch, int32_t relu,
                               int32_t act, void* stream_) {
  if (!dy || !y || !v || !gamma || !mean_rstd || !dv || !dgamma || !dbeta || !workspace || batch <= 0 || ch <= 0 || t <= 0) return TS_EINVAL;
  if (act < 0 || act > 1 || !rows_ok(dy, pitch, act) || !rows_ok(y, pitch, act) || !rows_ok(v, pitch, act) || !rows_ok(dv, pitch, act) || pitch < t) return TS_EINVAL;
  TS_STREAM;
  double* sums = static_cast<double*>(workspace);
  const dim3 rg = row_grid((long long)batch * ch, t);
  TS_ACT(act,
         { hipLaunchKernelGGL((chan_sums_kernel<1, float>), dim3(ch, BN_G), dim3(256), 0, stream, (const float*)dy, (const float*)y, (const float*)v, mean_rstd, sums,
                              batch, ch, t, pitch, relu);
           hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, rg, dim3(256), 0, stream, (const float*)dy, (const float*)y, (const float*)v, sums, gamma, mean_rstd,
                              (float*)dv, dgamma, dbeta, batch, ch, t, pitch, relu); },
         { hipLaunchKernelGGL((chan_sums_kernel<1, bf16_t>), dim3(ch, BN_G), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)v, mean_rstd, sums,
                              batch, ch, t, pitch, relu);
           hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, rg, dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)v, sums, gamma, mean_rstd,
                              (bf16_t*)dv, dgamma, dbeta, batch, ch, t, pitch, relu); });
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_add_relu_fwd(const void* a, const void* b, void* out, int64_t rows, int32_t t, int32_t pitch, int32_t act, void* stream_) {
  if (!a || !out || rows <= 0 || t <= 0 || pitch < t || act < 0 || act > 1) return TS_EINVAL;
  if (!rows_ok(a, pitch, act) || !rows_ok(out, pitch, act) || (b && !rows_ok(b, pitch, act))) return TS_EINVAL;
  TS_STREAM;
  TS_ACT(act,
         hipLaunchKernelGGL(add_relu_fwd_kernel<float>, row_grid(rows, t), dim3(256), 0, stream, (const float*)a, (const float*)b, (float*)out, (long long)rows, t, pitch),
         hipLaunchKernelGGL(add_relu_fwd_kernel<bf16_t>, row_grid(rows, t), dim3(256), 0, stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, (long long)rows, t, pitch));
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_add(const void* a, const void* b, const int32_t* len_b, int32_t ch, void* out, int64_t rows, int32_t t, int32_t pitch,
                            int32_t act, void* stream_) {
  if (!a || !b || !out || rows <= 0 || t <= 0 || pitch < t || act < 0 || act > 1 || (len_b && (ch <= 0 || rows % ch))) return TS_EINVAL;
  if (!rows_ok(a, pitch, act) || !rows_ok(out, pitch, act) || !rows_ok(b, pitch, act)) return TS_EINVAL;
  TS_STREAM;
  TS_ACT(act,
         hipLaunchKernelGGL(add_fwd_kernel<float>, row_grid(rows, t), dim3(256), 0, stream, (const float*)a, (const float*)b, (float*)out, (long long)rows, t, pitch, len_b, ch),
         hipLaunchKernelGGL(add_fwd_kernel<bf16_t>, row_grid(rows, t), dim3(256), 0, stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, (long long)rows, t, pitch, len_b, ch));
  return hip_status(hipGetLastError());
}

extern "C" int ts_train_relu_bwd(const void* dout, const void* out, void* din, int64_t rows, int32_t t, int32_t pitch, int32_t act, void* stream_) {
  if (!dout || !out || !din || rows <= 0 || t <= 0 || pitch < t || act < 0 || act > 1) return TS_EINVAL;
  if (!rows_ok(dout, pitch, act) || !rows_ok(out, pitch, act) || !rows_ok(din, pitch, act)) return TS_EINVAL;
  TS_STREAM;
  TS_ACT(act,
         hipLaunchKernelGGL(relu_bwd_kernel<float>, row_grid(rows, t), dim3(256), 0, stream, (const float*)dout, (const float*)out, (float*)din, (long long)rows, t, pitch),
         hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, row_grid(rows, t), dim3(256), 0, stream, (const bf16_t*)dout, (const bf16_t*)out, (bf16_t*)din, (long long)rows, t, pitch));
  return hip_status(hipGetLastError());
}
