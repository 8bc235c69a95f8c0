// Mel-filterbank front end for gfx950: pre-emphasis -> reflect-padded STFT power -> slaney mel -> log
// -> per-(clip, mel) masked normalisation, writing the bf16 NCT-p feature tensor the TCS stack consumes.
//
// Replaces FilterbankFeatures.forward in eval mode (reference: quartznet/transform.py:136-144 pre-emphasis,
// :186-208 torch.stft(center=True, reflect) + |.|^2, :243-255 mel matmul + log(x + 2^-24), :77-92 ->
// blocks.py:136-149 masked normalisation incl. quirk A1).  The reference materialises the complex
// spectrum (197 MB for 64 x 15 s) and six more full passes; here the waveform is read once, everything up
// to the log-mel stays in LDS/registers, and a second small kernel applies the per-row statistics.
//
// Kernel 1 (stft_mel_kernel): one workgroup = 16 frames, 16 lanes per frame.  A 512-point real FFT is done
// as a 256-point complex FFT of z[n] = x[2n] + i x[2n+1] by the four-step method (16 x 16): each lane runs
// a 16-point FFT in registers, the 16 lanes of a frame exchange through a padded LDS transpose, second
// 16-point FFT, then the real-FFT split, |X|^2, sparse triangular mel filters (CSR) and log.
// Workgroups are persistent (3 per CU: 52 KB of LDS each): tables are built once, the samples of the next frame group
// travel global -> registers while the current group is transformed.  The kernel is latency / barrier bound, not
// VALU bound (packed fp32 math changed nothing; a third resident workgroup per CU did: 121 -> 99 us).
// Kernel 2 (normalize_kernel): reduces the per-workgroup partial sums in fp64, normalises, masks frames
// >= length, transposes [frame][mel] -> [mel][frame] through LDS and stores bf16 rows.
#include "ts_common.hpp"
#include "ts_philox.hpp"

#include <cstdlib>

namespace ts {

constexpr int FPW = 16;            // frames per workgroup
constexpr int NFFT = 512;
constexpr int NC = NFFT / 2;       // complex FFT length
constexpr float LOG_FLOOR = 5.9604644775390625e-08f;   // 2^-24

struct Cx { float r, i; };
// Complex values live in two-float vectors so that hipcc emits the packed fp32 ops (v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32: one instruction per complex add, two per complex multiply) -- the kernel is VALU-issue bound.
using C2 = f32x2;
__device__ __forceinline__ C2 cmul(C2 a, C2 w) { return C2{a[0], a[0]} * w + C2{a[1], a[1]} * C2{-w[1], w[0]}; }
__device__ __forceinline__ C2 mul_mi(C2 a) { return C2{a[1], -a[0]}; }     // a * (-i)
__device__ __forceinline__ C2 mul_pi(C2 a) { return C2{-a[1], a[0]}; }     // a * (+i)

// 16-point forward DFT in registers (decimation 4 x 4), natural-order in and out.
__device__ __forceinline__ void fft16(C2 (&a)[16]) {
  C2 b[16];
  // stage 1: 4-point DFTs over n1 for each n2 (input index 4 n1 + n2), result index [n2][k1]
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) {
    const C2 s02 = a[n2] + a[8 + n2], d02 = a[n2] - a[8 + n2];
    const C2 s13 = a[4 + n2] + a[12 + n2], d13 = a[4 + n2] - a[12 + n2];
    b[n2 * 4 + 0] = s02 + s13;
    b[n2 * 4 + 1] = d02 + mul_mi(d13);             // x0 - i x1 - x2 + i x3
    b[n2 * 4 + 2] = s02 - s13;
    b[n2 * 4 + 3] = d02 + mul_pi(d13);             // x0 + i x1 - x2 - i x3
  }
  // twiddles W16^(n2 k1)
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
  const C2 w[10] = {C2{1.f, 0.f}, C2{C1, -S1}, C2{R2, -R2}, C2{S1, -C1}, C2{0.f, -1.f},
                    C2{-S1, -C1}, C2{-R2, -R2}, C2{-C1, -S1}, C2{-1.f, 0.f}, C2{-C1, S1}};
#pragma unroll
  for (int n2 = 1; n2 < 4; ++n2)
#pragma unroll
    for (int k1 = 1; k1 < 4; ++k1) b[n2 * 4 + k1] = cmul(b[n2 * 4 + k1], w[n2 * k1]);
  // stage 2: 4-point DFTs over n2 for each k1, output index k1 + 4 k2
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    const C2 s02 = b[k1] + b[8 + k1], d02 = b[k1] - b[8 + k1];
    const C2 s13 = b[4 + k1] + b[12 + k1], d13 = b[4 + k1] - b[12 + k1];
    a[k1 + 0] = s02 + s13;
    a[k1 + 4] = d02 + mul_mi(d13);
    a[k1 + 8] = s02 - s13;
    a[k1 + 12] = d02 + mul_pi(d13);
  }
}

// A frame's 16 lanes live in ONE wave (4 frames per wave) and its scratch tile is private to them, so the exchanges inside a frame need no
// workgroup barrier: LDS operations of a wave complete in issue order; this only stops the compiler from moving accesses across the hand-over.
// (Round 5: five of the seven workgroup barriers per frame group became this; front end 121.5 -> 113 us on the same box.  A per-clip finalize
// kernel for the normaliser's statistics -- instead of every normalize workgroup re-reducing the partial sums -- measured neutral and is not here.)
__device__ __forceinline__ void frame_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct FeArgs {
  const float* wave;
  const int* wave_len;
  const float* window;        // [NFFT]
  const float* mel_w;         // CSR weights
  const int* mel_off;         // [n_mels + 1][2] = (first bin, offset)
  float* logmel;              // [B][F][n_mels]
  float* partial;             // [B][nwg][n_mels][2]
  int* feat_len;
  long long* feat_len64;      // optional int64 copy of feat_len (the lengths the Python side returns)
  int n_samples, hop, n_mels, n_frames, nwg, mel_nnz, batch;
  float preemph;
  float dither;               // > 0: DitherAudio (transform.py:109-118, training only): x + dither * N(0, 1)
  unsigned long long seed;    // Philox key of the dither noise
};

// Dither noise of sample `k` of clip `b`: a pure function of (seed, b, k), so every frame group (and the reflect padding)
// that touches the sample sees the same value -- exactly as if the noise had been added to the waveform up front.
__device__ __forceinline__ float dither_noise(unsigned long long seed, int b, int k) {
  const Philox4 r = philox(seed, PHILOX_DITHER, ((unsigned long long)(unsigned)b << 32) | (unsigned)(k >> 1));
  float n0, n1;
  normal2(r.v[0], r.v[1], n0, n1);
  return (k & 1) ? n1 : n0;
}

// DITHER is a template flag: the Philox code, unrolled into the staging of every sample, tripled the kernel's size (and spilled 49 SGPRs) for a
// branch that eval mode never takes.
template <bool DITHER>
__global__ __launch_bounds__(256) void stft_mel_kernel(const FeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int span = (FPW - 1) * a.hop + NFFT;
  float* const sig = reinterpret_cast<float*>(smem);                 // [span] pre-emphasised, reflect-padded
  Cx* const tw256 = reinterpret_cast<Cx*>(sig + round_up(span, 4));  // W_256^j
  float* const win = reinterpret_cast<float*>(tw256 + NC);           // [NFFT]
  float* const fr = win + NFFT;                                      // per frame scratch
  constexpr int YSZ = 16 * 17 * 2;                                   // floats: padded transpose tile; Z and P reuse it
  constexpr int FSZ = YSZ;
  static_assert(NC * 2 <= YSZ, "the spectrum of a frame fits the transpose tile it replaces");
  constexpr int RED0 = 288;                                          // log-mel staging of a frame: floats [288, 288 + n_mels) of its
                                                                     // tile, above the 257 power-spectrum bins (n_mels <= 256)
  float* const melw = fr + FPW * FSZ;                                // [mel_nnz] CSR weights
  int* const melo = reinterpret_cast<int*>(melw + a.mel_nnz);        // [n_mels + 1][2]

  const int tid = threadIdx.x;
  const int T = a.n_samples;

  // ---- tables: once per (persistent) workgroup -------------------------------------------------------------
  for (int j = tid; j < NC; j += 256) {
    float s, c;
    sincospif(-2.0f * (float)j / (float)NC, &s, &c);
    tw256[j] = Cx{c, s};
  }
  for (int j = tid; j < NFFT; j += 256) win[j] = a.window[j];
  for (int j = tid; j < a.mel_nnz; j += 256) melw[j] = a.mel_w[j];
  for (int j = tid; j < (a.n_mels + 1) * 2; j += 256) melo[j] = a.mel_off[j];

  // ---- signal staging, one frame group ahead: the samples of group g + grid travel global -> registers while group g
  // is transformed, and go into `sig` as soon as the first FFT stage of g has read it.  Interior groups (no reflection, no
  // clip end inside the span: all but two or three per clip) take 16-byte loads -- three per thread plus the sample in front
  // of each for the pre-emphasis -- and 16-byte LDS writes; the others are staged element by element when their turn comes. ----
  constexpr int NV = 3;                              // 16-byte groups per thread held in registers (span <= 3072)
  const int span4 = round_up(span, 4) / 4;
  const bool vec_ok = span4 <= 256 * NV && (T & 3) == 0 && ((FPW * a.hop) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.wave) & 15) == 0;
  auto interior = [&](int g) {
    const int k0 = (g % a.nwg) * FPW * a.hop - NFFT / 2;
    return vec_ok && k0 >= 4 && k0 + 4 * span4 <= T;
  };
  f32x4 cur[NV];
  float prv[NV];
  auto fetch = [&](int g) {
    const float* x = a.wave + (size_t)(g / a.nwg) * a.n_samples + ((g % a.nwg) * FPW * a.hop - NFFT / 2);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int e = tid + 256 * j < span4 ? tid + 256 * j : span4 - 1;      // unconditional loads (a guarded load is waited for on the spot)
      // asm loads: hipcc arranges the loaded registers for the commit right behind a plain load -- and waits for it there, which makes the prefetch synchronous
      const float* const src = x + 4 * e;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(cur[j]) : "v"(src) : "memory");
      asm volatile("global_load_dword %0, %1, off offset:-4" : "=v"(prv[j]) : "v"(src) : "memory");
    }
  };
  auto commit = [&](int g) {
    // the asm loads of fetch() (hipcc does not count them); the registers are operands of the wait so that no use of them moves above it
    static_assert(NV == 3, "operand list of the wait below");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(prv[0]), "+v"(prv[1]), "+v"(prv[2]) :: "memory");
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int e = tid + 256 * j;
      if constexpr (DITHER) {                          // only now: the registers are not to be touched before the wait above
        const int b = g / a.nwg, k = (g % a.nwg) * FPW * a.hop - NFFT / 2 + 4 * (e < span4 ? e : span4 - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[j][q] += a.dither * dither_noise(a.seed, b, k + q);
        prv[j] += a.dither * dither_noise(a.seed, b, k - 1);
      }
      if (e < span4)
        *reinterpret_cast<f32x4*>(sig + 4 * e) = f32x4{cur[j][0] - a.preemph * prv[j], cur[j][1] - a.preemph * cur[j][0],
                                                       cur[j][2] - a.preemph * cur[j][1], cur[j][3] - a.preemph * cur[j][2]};
    }
  };
  auto stage_direct = [&](int g) {                   // edge groups and long hops: no register prefetch
    const float* x = a.wave + (size_t)(g / a.nwg) * a.n_samples;
    const int k0 = (g % a.nwg) * FPW * a.hop - NFFT / 2;
    for (int e = tid; e < span; e += 256) {
      int k = k0 + e;
      k = k < 0 ? -k : k;
      k = k >= T ? 2 * (T - 1) - k : k;
      k = k < 0 ? 0 : (k >= T ? T - 1 : k);          // frames entirely beyond the clip (never valid)
      float xc = x[k], xp = k > 0 ? x[k - 1] : 0.f;  // sample 0 is not pre-emphasised
      if constexpr (DITHER) {
        xc += a.dither * dither_noise(a.seed, g / a.nwg, k);
        if (k > 0) xp += a.dither * dither_noise(a.seed, g / a.nwg, k - 1);
      }
      sig[e] = xc - a.preemph * xp;
    }
  };
  const int n_groups = a.nwg * a.batch;
  int g = blockIdx.x;
  if (interior(g)) { fetch(g); commit(g); } else { stage_direct(g); }
  __syncthreads();

  for (; g < n_groups; g += gridDim.x) {
  const int b = g / a.nwg, grp = g - b * a.nwg;
  const int f0 = grp * FPW;
  const int gn = g + gridDim.x;
  const bool nvec = gn < n_groups && interior(gn);
  if (nvec) fetch(gn);

  // ---- 256-point complex FFT per frame: 16 lanes per frame ---------------------------------------------
  const int fl = tid >> 4;          // frame in workgroup
  const int i = tid & 15;
  float* const Y = fr + fl * FSZ;
  float* const Z = Y;                 // written only after every lane of the workgroup has read its Y row
  {
    C2 v[16];
    const float* s = sig + fl * a.hop;
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const int j = 32 * n1 + 2 * i;
      v[n1] = *reinterpret_cast<const f32x2*>(s + j) * *reinterpret_cast<const f32x2*>(win + j);
    }
    fft16(v);                                    // over n1 -> index k1
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1)
      *reinterpret_cast<f32x2*>(Y + (k1 * 17 + i) * 2) = cmul(v[k1], *reinterpret_cast<const f32x2*>(&tw256[(i * k1) & (NC - 1)]));
  }
  frame_sync();
  {
    C2 v[16];
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) v[n2] = *reinterpret_cast<const f32x2*>(Y + (i * 17 + n2) * 2);
    frame_sync();
    fft16(v);                                    // over n2 -> k2; bin = i + 16 k2
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) *reinterpret_cast<f32x2*>(Z + (i + 16 * k2) * 2) = v[k2];
  }
  frame_sync();
  // ---- real-FFT split + power spectrum, bins k = i + 16 j (and bin 256 on lane 0) -> P (aliases Y) ------
  float* const P = Y;
  {
    float pw[17];
    const f32x2 w_odd = (i & 1) ? f32x2{0.99992470183914454f, -0.012271538285719925f} : f32x2{1.f, 0.f};   // W_512^1
#pragma unroll
    for (int j = 0; j < 17; ++j) {
      const int k = i + 16 * j;
      if (j < 16 || i == 0) {
        const f32x2 zk = *reinterpret_cast<const f32x2*>(Z + ((k & (NC - 1)) * 2));
        const f32x2 zn = *reinterpret_cast<const f32x2*>(Z + (((NC - k) & (NC - 1)) * 2));
        // W_512^k = W_256^(k >> 1) * W_512^(k & 1); k = i + 16 j has the parity of the lane
        const f32x2 w = cmul(*reinterpret_cast<const f32x2*>(&tw256[(k >> 1) & (NC - 1)]), w_odd);
        const f32x2 znc = f32x2{zn[0], -zn[1]};                          // conj(z[N - k])
        const f32x2 A = zk + znc, D = zk - znc;                          // A = (ar, ai), D = (dr, di)
        // X = 0.5 * (A - i W D):  -i (W D) = (Im(WD), -Re(WD))
        const f32x2 wd = cmul(D, w);
        const f32x2 X = 0.5f * (A + f32x2{wd[1], -wd[0]});
        const float xr = X[0], xi = X[1];
        pw[j] = xr * xr + xi * xi;
      }
    }
    frame_sync();                                 // all lanes of the frame are done reading Y/Z rows
#pragma unroll
    for (int j = 0; j < 17; ++j)
      if (j < 16 || i == 0) P[i + 16 * j] = pw[j];
  }
  frame_sync();
  // ---- sparse mel filters + log -------------------------------------------------------------------------
  const int f = f0 + fl;
  for (int m = i; m < a.n_mels; m += 16) {
    const int first = melo[2 * m], off = melo[2 * m + 1];
    const int cnt = melo[2 * m + 3] - off;
    float acc = 0.f;
    for (int j = 0; j < cnt; ++j) acc = fmaf(melw[off + j], P[first + j], acc);
    const float lm = logf(acc + LOG_FLOOR);
    fr[fl * FSZ + RED0 + m] = lm;
    if (f < a.n_frames) a.logmel[((size_t)b * a.n_frames + f) * a.n_mels + m] = lm;
  }
  __syncthreads();                                   // every wave has long read its samples of group g and staged its frames' log-mel
  if (nvec) commit(gn);                              // the next group's samples (prefetched into registers above)
  // ---- partial statistics over the valid frames of this workgroup -----------------------------------------
  const int flen = a.wave_len[b] / a.hop + 1;      // floor(len / hop) + 1  (transform.py:182-184)
  if (tid == 0 && grp == 0) {
    a.feat_len[b] = flen;
    if (a.feat_len64) a.feat_len64[b] = flen;
  }
  for (int m = tid; m < a.n_mels; m += 256) {
    float s1 = 0.f, s2 = 0.f;
    for (int q = 0; q < FPW; ++q) {
      if (f0 + q < flen && f0 + q < a.n_frames) {
        const float v = fr[q * FSZ + RED0 + m];
        s1 += v;
        s2 = fmaf(v, v, s2);
      }
    }
    float* dst = a.partial + (((size_t)b * a.nwg + grp) * a.n_mels + m) * 2;
    dst[0] = s1;
    dst[1] = s2;
  }
  if (!nvec && gn < n_groups) stage_direct(gn);
  __syncthreads();                                   // `red` / P are free again, the next group's samples are in place
  }
}

struct NormArgs {
  const float* logmel;        // [B][F][n_mels]
  const float* partial;       // [B][nwg][n_mels][2]
  const int* feat_len;
  unsigned short* out;        // [B][n_mels][pitch]
  int n_mels, n_frames, nwg, pitch;
  const int* masks;           // SpecAugment / SpecCutout rectangles [n_masks][4] = (f0, f1, t0, t1), or NULL
  int n_masks;
};

constexpr int NTF = 64;            // frames per normalize workgroup

__global__ __launch_bounds__(256) void normalize_kernel(const NormArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const mean = reinterpret_cast<float*>(smem);           // [n_mels]
  float* const rstd = mean + a.n_mels;                          // [n_mels]
  float* const tile = rstd + a.n_mels;                          // [NTF][n_mels + 1]
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * NTF;
  const int flen = a.feat_len[b] < a.n_frames ? a.feat_len[b] : a.n_frames;

  // statistics: 4 interleaved slices of the per-workgroup partial sums per mel row (all 256 threads busy), combined in a
  // fixed order -> deterministic; the slice sums borrow the transpose tile before it is filled
  constexpr int NSL = 4;
  double* const slice = reinterpret_cast<double*>(tile);        // [NSL][n_mels][2]
  for (int idx = tid; idx < NSL * a.n_mels; idx += 256) {
    const int sl = idx / a.n_mels, m = idx - sl * a.n_mels;
    double s1 = 0.0, s2 = 0.0;
    const float* p = a.partial + ((size_t)b * a.nwg * a.n_mels + m) * 2;
    for (int w = sl; w < a.nwg; w += NSL) {
      s1 += (double)p[(size_t)w * a.n_mels * 2];
      s2 += (double)p[(size_t)w * a.n_mels * 2 + 1];
    }
    slice[idx * 2] = s1;
    slice[idx * 2 + 1] = s2;
  }
  __syncthreads();
  for (int m = tid; m < a.n_mels; m += 256) {
    double s1 = 0.0, s2 = 0.0;
    for (int sl = 0; sl < NSL; ++sl) {
      s1 += slice[(sl * a.n_mels + m) * 2];
      s2 += slice[(sl * a.n_mels + m) * 2 + 1];
    }
    const double n = (double)flen;
    const double mu = s1 / n;
    // quirk A1: padded frames contribute mu^2 each to the variance numerator
    double var = (s2 - n * mu * mu + (double)(a.n_frames - flen) * mu * mu) / n;
    var = var < 0.0 ? 0.0 : var;
    mean[m] = (float)mu;
    rstd[m] = (float)(1.0 / (sqrt(var) + 1e-5));
  }
  __syncthreads();
  const int ld = a.n_mels + 1;
  for (int idx = tid; idx < NTF * a.n_mels; idx += 256) {
    const int q = idx / a.n_mels, m = idx - q * a.n_mels;
    const int f = f0 + q;
    tile[q * ld + m] = f < a.n_frames ? a.logmel[((size_t)b * a.n_frames + f) * a.n_mels + m] : 0.f;
  }
  __syncthreads();
  // each thread produces 8 consecutive frames of one mel row (16 B)
  for (int idx = tid; idx < a.n_mels * (NTF / 8); idx += 256) {
    const int m = idx / (NTF / 8), g = idx - m * (NTF / 8);
    const float mu = mean[m], rs = rstd[m];
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = f0 + g * 8 + j;
      v[j] = f < flen ? (tile[(g * 8 + j) * ld + m] - mu) * rs : 0.f;
    }
    // spec_augment.py:51-56, :96-101: masked_fill(mask, 0) after the normaliser -- applied on the way out
    for (int r = 0; r < a.n_masks; ++r) {
      const int mf0 = a.masks[4 * r], mf1 = a.masks[4 * r + 1], mt0 = a.masks[4 * r + 2], mt1 = a.masks[4 * r + 3];
      if (m >= mf0 && m < mf1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int f = f0 + g * 8 + j;
          if (f >= mt0 && f < mt1) v[j] = 0.f;
        }
      }
    }
    if (f0 + g * 8 < a.pitch)
      *reinterpret_cast<u32x4*>(a.out + ((size_t)b * a.n_mels + m) * a.pitch + f0 + g * 8) =
          u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
  }
}

static inline int fe_nwg(const ts_frontend_desc* d) { return (d->n_frames + FPW - 1) / FPW; }
static inline size_t fe_logmel_bytes(const ts_frontend_desc* d) {
  return round_up((size_t)d->batch * d->n_frames * d->n_mels * sizeof(float), 256);
}

}  // namespace ts

extern "C" int64_t ts_frontend_workspace_bytes(const ts_frontend_desc* d) {
  if (!d || d->batch <= 0 || d->n_frames <= 0 || d->n_mels <= 0) return TS_EINVAL;
  return (int64_t)(ts::fe_logmel_bytes(d) + (size_t)d->batch * ts::fe_nwg(d) * d->n_mels * 2 * sizeof(float));
}

extern "C" const float* ts_frontend_logmel_ptr(const ts_frontend_desc* d, const void* workspace) {
  (void)d;
  return static_cast<const float*>(workspace);
}

extern "C" int ts_mel_frontend_fwd(const ts_frontend_desc* d, const float* wave, const int32_t* wave_len, void* features,
                                   int32_t* feat_len, void* workspace, void* stream_) {
  using namespace ts;
  if (!d || !wave || !wave_len || !features || !feat_len || !workspace) return TS_EINVAL;
  if (!d->window || !d->mel_weights || !d->mel_offsets) return TS_EINVAL;
  if (d->batch <= 0 || d->n_samples <= NFFT / 2 || d->hop <= 0 || d->n_mels <= 0) return TS_EINVAL;
  if (d->n_fft != NFFT) return TS_EUNSUPPORTED;                    // both model families use n_fft = 512
  if (d->win_length > NFFT || d->win_length <= 0) return TS_EINVAL;  // transform.py:166-170
  if (d->n_frames != d->n_samples / d->hop + 1) return TS_EINVAL;
  if (d->pitch_out % 8 || d->pitch_out < d->n_frames) return TS_EINVAL;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int nwg = fe_nwg(d);

  FeArgs a{};
  a.wave = wave; a.wave_len = wave_len; a.window = d->window; a.mel_w = d->mel_weights; a.mel_off = d->mel_offsets;
  a.logmel = static_cast<float*>(workspace);
  a.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + fe_logmel_bytes(d));
  a.feat_len = feat_len;
  a.feat_len64 = reinterpret_cast<long long*>(d->feat_len64);
  a.n_samples = d->n_samples; a.hop = d->hop; a.n_mels = d->n_mels; a.n_frames = d->n_frames; a.nwg = nwg;
  a.preemph = d->preemph;
  a.dither = d->dither;
  a.seed = d->dither_seed;
  if (d->n_masks < 0 || (d->n_masks > 0 && !d->masks)) return TS_EINVAL;
  a.mel_nnz = d->mel_nnz;
  a.batch = d->batch;
  const int span = (FPW - 1) * d->hop + NFFT;
  if (d->mel_nnz < 0) return TS_EINVAL;
  if (d->n_mels > 256) return TS_EUNSUPPORTED;                     // the log-mel staging sits in the upper half of a frame tile
  const size_t lds1 = ((size_t)round_up(span, 4) + NC * 2 + NFFT + (size_t)FPW * (16 * 17 * 2) +
                       (size_t)d->mel_nnz + (size_t)(d->n_mels + 1) * 2) * sizeof(float);
  if (lds1 > 160 * 1024) return TS_EUNSUPPORTED;
  if (lds1 > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stft_mel_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(stft_mel_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    if (e != hipSuccess) return (int)e;
  }
  (void)hipGetLastError();
  // persistent workgroups: as many as are resident at once, each walks frame groups g, g + grid, ...  (LDS allows three per CU; the dither instantiation's
  // registers only two -- a third would start when the first two have finished their share)
  int per_cu = (int)((160 * 1024) / lds1) < 1 ? 1 : (int)((160 * 1024) / lds1);
  {
    static int occ_cache[64][2] = {};          // per (device, instantiation); the query depends on lds1 too, which the two model families share
    static size_t occ_lds[64][2] = {};
    int dev = 0;
    const int di = a.dither > 0.f ? 1 : 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
      if (occ_lds[dev][di] != lds1) {
        int occ = 0;
        const hipError_t oe = di ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, stft_mel_kernel<true>, 256, lds1)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, stft_mel_kernel<false>, 256, lds1);
        (void)hipGetLastError();
        occ_cache[dev][di] = (oe == hipSuccess && occ > 0) ? occ : per_cu;
        occ_lds[dev][di] = lds1;
      }
      if (occ_cache[dev][di] < per_cu) per_cu = occ_cache[dev][di];
    }
  }
  const int n_groups = nwg * d->batch;
  const int grid = n_groups < cu_count() * per_cu ? n_groups : cu_count() * per_cu;
  if (a.dither > 0.f) hipLaunchKernelGGL(stft_mel_kernel<true>, dim3(grid), dim3(256), lds1, stream, a);
  else hipLaunchKernelGGL(stft_mel_kernel<false>, dim3(grid), dim3(256), lds1, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;

  NormArgs n{};
  n.logmel = a.logmel; n.partial = a.partial; n.feat_len = feat_len; n.out = static_cast<unsigned short*>(features);
  n.n_mels = d->n_mels; n.n_frames = d->n_frames; n.nwg = nwg; n.pitch = d->pitch_out;
  n.masks = d->masks; n.n_masks = d->n_masks;
  const size_t lds2 = ((size_t)2 * d->n_mels + (size_t)NTF * (d->n_mels + 1)) * sizeof(float);
  (void)hipGetLastError();
  hipLaunchKernelGGL(normalize_kernel, dim3((d->pitch_out + NTF - 1) / NTF, d->batch), dim3(256), lds2, stream, n);
  return hip_status(hipGetLastError());
}
