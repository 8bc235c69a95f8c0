// Fused time-channel-separable sub-block for gfx950 (MI355X), inference.
//
//   y[b, co, t] = act( sum_ci Wf[co, ci] * mask(dw[b, ci, t]) + bias[co] + sum_cr Wr[co, cr] * mask(xres[b, cr, t*rs]) )
//   dw[b, ci, t] = sum_u taps[ci, u] * mask(x)[b, ci, t*stride + u*dil - pad]
//
// Replaces the reference's per-sub-block ATen chain (quartznet/blocks.py:166-182 masked_fill + conv1d
// (groups=C) + masked_fill + conv1d(k=1), :222 batch_norm, :332-337 residual add + relu).
//
// Design (see DESIGN.md "TCS kernel"):
//  * layout NCT-p: bf16 [B][C][Tp], time contiguous.  One workgroup owns a tile of TT output frames x
//    CO_WG output channels of one clip and walks the input channels in stages of 64.
//  * 8 waves with fixed roles.  Waves 4-7 are PRODUCERS: each stages 16 input channels of the stage
//    (global -> registers -> wave-private LDS rows, one stage ahead), runs the depthwise FIR and writes the
//    bf16 result tile dwt[stage & 1].  Waves 0-3 are CONSUMERS: they hold the fp32 accumulators and, one
//    stage behind the producers, run the pointwise GEMM out of dwt.  One s_barrier per stage; the two
//    roles overlap memory latency, LDS traffic and matrix work of adjacent stages.
//  * depthwise FIR on the matrix cores: v_mfma_f32_4x4x4_16b_bf16 computes 16 independent 4x4x4
//    products per instruction -- one block per channel.  For channel c the A block is a 4x4 slice of the
//    Toeplitz matrix of its taps (rows = 4 consecutive output frames; pre-shifted per row on the host),
//    the B block holds 4 consecutive input samples for each of 4 time runs.  A lane keeps its sliding
//    input window in registers.  4.1x the fp32-VALU FMA rate measured on MI355X (240 vs 58 TMAC/s).
//  * the depthwise result is stored [ci][t] (XOR-swizzled 16-B chunks) and consumed as the A operand of
//    v_mfma_f32_32x32x16_bf16 through ds_read_b64_tr_b16 (hardware transpose read); the B operand
//    (BN-folded pointwise weights) is pre-packed per lane on the host and streamed from L2 through a
//    2-deep register ring.
//  * residual 1x1 conv = extra stages over the block input whose "depthwise" is a masked copy,
//    accumulating into the same registers; bias + ReLU + bf16 pack in the epilogue.
#include "ts_common.hpp"

namespace ts {

constexpr int KC = 64;         // input channels per stage
constexpr int NKP = 3;         // depthwise k-steps (of 4 samples) per pass
constexpr int XMAX = 5;        // staged row length <= 64 * XMAX elements
constexpr int NKMAX = 24;      // taps are cached in LDS up to this many k-steps

struct TcsArgs {
  const unsigned short* x;     // [B][c_in][pitch_in]
  const unsigned short* xres;  // [B][c_res][pitch_res]
  void* y;                     // [B][c_out][pitch_out] bf16 or f32
  const int* len_in;
  const int* len_res;
  const unsigned short* taps;  // [c_in_pad][4][4*nk]
  const unsigned short* pw_w;  // fragments
  const unsigned short* res_w;
  const float* bias;
  int c_in, c_out, c_res;
  int pitch_in, pitch_out, pitch_res;
  int t_out;
  int kernel, stride, dilation, padding;
  int npass;                   // nk = 3 * npass
  int woff;                    // padL8 - padL4: element offset of the lane windows inside an xs row
  int padl8;                   // xs row starts at input frame t0*stride - padl8
  int xe;                      // staged elements per xs row (multiple of 8)
  int xpitch;                  // xs row pitch in elements (multiple of 4)
  int relu;
  int res_stride;
  int kt_main, kt_res;         // k-steps (16 channels) in the packed weights = c_pad64 / 16
  int taps_lds;                // 1: taps of the stage are cached in LDS
};

// [ci][t] bf16 tile of the depthwise output / identity input, 16-byte chunks XOR-swizzled so that both
// the 8-byte row writes and the transposed reads spread over the banks.
template <int TT>
struct DwTile {
  static constexpr int ROWB = TT * 2;
  static constexpr int BYTES = KC * ROWB;
  __device__ static __forceinline__ int sw(int c) {
    return TT == 128 ? (c & 3) * 5 : ((((c >> 1) & 1) << 2) | (c & 3));
  }
  __device__ static __forceinline__ int addr(int c, int t) {
    return c * ROWB + ((((t >> 3) ^ sw(c))) << 4) + ((t & 7) << 1);
  }
};

__device__ __forceinline__ int conv_len(int len, int k, int s, int p, int d) {
  const int num = len + 2 * p - d * (k - 1) - 1;
  return num < 0 ? 0 : num / s + 1;
}

// LDS writes of this wave complete -> workgroup barrier.  Outstanding global loads stay in flight.
__device__ __forceinline__ void stage_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int TT, int NT, int STRIDE, bool DW, bool OUT_F32>
__global__ __launch_bounds__(512, 2) void tcs_kernel(const TcsArgs a) {
  constexpr int MT = TT / 32;     // 32-frame MFMA row tiles
  constexpr int M = TT / 16;      // 4-frame steps per lane run (4 runs per channel)
  constexpr int RUN = TT / 4;
  constexpr int NPP = (M - 1) * STRIDE + NKP;   // window pairs live per pass
  constexpr int IDJ = TT / 64;    // identity staging: 16-B column groups per lane per row
  using Tile = DwTile<TT>;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const dwt = smem;                          // [2][KC][TT] bf16 swizzled
  char* const xs = smem + 2 * Tile::BYTES;         // [4 producers][16][xpitch] bf16 (DW only)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * TT;

  const int n_main = (a.c_in + KC - 1) / KC;
  const int n_res = a.c_res > 0 ? (a.c_res + KC - 1) / KC : 0;
  const int n_stage = n_main + n_res;

  if (wave >= 4) {
    // ================================= PRODUCER =======================================================
    const int pv = wave - 4;                     // producer index: channels [16 pv, 16 pv + 16) of the stage
    const int r8 = lane >> 3, sub = lane & 7;    // staging: rows r8, r8 + 8; 16-B column groups sub + 8j
    const int len_in = a.len_in[b];
    const int len_res = a.c_res > 0 ? a.len_res[b] : 0;

    // ---- identity staging (pointwise-only main source, residual source) ------------------------------
    u32x4 I[2][IDJ];
    auto id_src = [&](int s, const unsigned short*& src, int& c_total, int& pitch, int& len, int& rs, int& chunk) {
      if (!DW && s < n_main) { src = a.x; c_total = a.c_in; pitch = a.pitch_in; len = len_in; rs = STRIDE; chunk = s; }
      else { src = a.xres; c_total = a.c_res; pitch = a.pitch_res; len = len_res; rs = a.res_stride; chunk = s - n_main; }
    };
    auto issue_id = [&](int s) {
      const unsigned short* src; int c_total, pitch, len, rs, chunk;
      id_src(s, src, c_total, pitch, len, rs, chunk);
      if (rs != 1) return;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int c = chunk * KC + pv * 16 + r8 + 8 * rr;
#pragma unroll
        for (int j = 0; j < IDJ; ++j) {
          const int t = t0 + (sub + 8 * j) * 8;
          I[rr][j] = u32x4{0u, 0u, 0u, 0u};
          if (c < c_total && t < pitch)
            I[rr][j] = *reinterpret_cast<const u32x4*>(src + ((size_t)(b * c_total + c) * pitch + t));
        }
      }
    };
    auto write_id = [&](int s, char* dst) {
      const unsigned short* src; int c_total, pitch, len, rs, chunk;
      id_src(s, src, c_total, pitch, len, rs, chunk);
      if (rs == 1) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int cl = pv * 16 + r8 + 8 * rr;
#pragma unroll
          for (int j = 0; j < IDJ; ++j) {
            const int tl = (sub + 8 * j) * 8;
            *reinterpret_cast<u32x4*>(dst + Tile::addr(cl, tl)) = keep_first(I[rr][j], len - (t0 + tl));
          }
        }
      } else {
        // strided 1x1 (Citrinet strided residual, generic strided pointwise): plain gather
        for (int idx = lane; idx < 16 * TT; idx += 64) {
          const int cl = pv * 16 + idx / TT, tl = idx % TT;
          const int c = chunk * KC + cl, ti = (t0 + tl) * rs;
          unsigned short v = 0;
          if (c < c_total && ti < len && ti < pitch) v = src[(size_t)(b * c_total + c) * pitch + ti];
          *reinterpret_cast<unsigned short*>(dst + Tile::addr(cl, tl)) = v;
        }
      }
    };

    if constexpr (DW) {
      const int len_mid = conv_len(len_in, a.kernel, STRIDE, a.padding, a.dilation);
      const int nk = a.npass * NKP;
      const int cw = pv * 16 + (lane >> 2);      // channel inside the stage handled by this lane
      const int q = lane & 3;                    // time run (B column) and Toeplitz row (A row)
      const int G = a.xe >> 3;
      const int tin0 = t0 * STRIDE - a.padl8;
      char* const xs_w = xs + (size_t)pv * 16 * a.xpitch * 2;
      char* const tl_w = xs + (size_t)4 * 16 * a.xpitch * 2 + (size_t)pv * NKMAX * 512;

      u32x4 X[2][XMAX];
      u32x2 T[NKMAX];
      auto issue_x = [&](int chunk) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int c = chunk * KC + pv * 16 + r8 + 8 * rr;
          const unsigned short* src = a.x + (size_t)(b * a.c_in + c) * a.pitch_in;
#pragma unroll
          for (int j = 0; j < XMAX; ++j) {
            const int g = sub + 8 * j;
            const int t = tin0 + 8 * g;
            X[rr][j] = u32x4{0u, 0u, 0u, 0u};
            if (g < G && c < a.c_in && t >= 0 && t < a.pitch_in) X[rr][j] = *reinterpret_cast<const u32x4*>(src + t);
          }
        }
      };
      auto write_x = [&]() {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int row = r8 + 8 * rr;
#pragma unroll
          for (int j = 0; j < XMAX; ++j) {
            const int g = sub + 8 * j;
            if (g < G) {
              const u32x4 v = keep_first(X[rr][j], len_in - (tin0 + 8 * g));
              u32x2* dst = reinterpret_cast<u32x2*>(xs_w + ((size_t)row * a.xpitch + g * 8) * 2);
              dst[0] = u32x2{v[0], v[1]};
              dst[1] = u32x2{v[2], v[3]};
            }
          }
        }
      };
      auto tap_ptr = [&](int chunk) {
        return a.taps + ((size_t)((chunk * KC + cw) * 4 + q) * nk) * 4;
      };
      auto issue_t = [&](int chunk) {
        if (!a.taps_lds) return;
        const unsigned short* tp = tap_ptr(chunk);
#pragma unroll
        for (int s = 0; s < NKMAX; ++s)
          if (s < nk) T[s] = *reinterpret_cast<const u32x2*>(tp + s * 4);
      };
      auto write_t = [&]() {
        if (!a.taps_lds) return;
#pragma unroll
        for (int s = 0; s < NKMAX; ++s)
          if (s < nk) *reinterpret_cast<u32x2*>(tl_w + (s * 64 + lane) * 8) = T[s];
      };

      issue_x(0);
      issue_t(0);
      for (int s = 0; s < n_main; ++s) {
        char* const dst = dwt + (s & 1) * Tile::BYTES;
        write_x();
        write_t();
        if (s + 1 < n_main) { issue_x(s + 1); issue_t(s + 1); }
        else if (s + 1 < n_stage) issue_id(s + 1);
        // ---- depthwise on v_mfma_f32_4x4x4_16b_bf16
        f32x4 d[M];
#pragma unroll
        for (int m = 0; m < M; ++m) d[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* xrow = xs_w + ((size_t)(cw - pv * 16) * a.xpitch + a.woff + q * RUN * STRIDE) * 2;
        const unsigned short* tp = tap_ptr(s);
        for (int pass = 0; pass < a.npass; ++pass) {
          s16x4 A[NKP];
#pragma unroll
          for (int k = 0; k < NKP; ++k) {
            if (a.taps_lds) A[k] = *reinterpret_cast<const s16x4*>(tl_w + ((pass * NKP + k) * 64 + lane) * 8);
            else A[k] = *reinterpret_cast<const s16x4*>(tp + (pass * NKP + k) * 4);
          }
          s16x4 P[NPP];
#pragma unroll
          for (int u = 0; u < NPP; ++u) P[u] = *reinterpret_cast<const s16x4*>(xrow + (pass * NKP + u) * 8);
#pragma unroll
          for (int k = 0; k < NKP; ++k)
#pragma unroll
            for (int m = 0; m < M; ++m)
              d[m] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(A[k], P[m * STRIDE + k], d[m], 0, 0, 0);
        }
        // mask frames >= len_mid (quirk A2: the pointwise conv sees a re-masked input) and store
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const int tl = q * RUN + 4 * m;
          const int t = t0 + tl;
          f32x4 v = d[m];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = (t + i < len_mid) ? v[i] : 0.f;
          *reinterpret_cast<u32x2*>(dst + Tile::addr(cw, tl)) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
        }
        stage_barrier();
      }
    } else {
      issue_id(0);
    }
    // identity stages: all stages when !DW, else the residual ones (first one already prefetched)
    for (int s = DW ? n_main : 0; s < n_stage; ++s) {
      write_id(s, dwt + (s & 1) * Tile::BYTES);
      if (s + 1 < n_stage) issue_id(s + 1);
      stage_barrier();
    }
    return;
  }

  // =================================== CONSUMER =======================================================
  const int cot0 = (blockIdx.z * 4 + wave) * NT;   // first 32-channel output tile of this wave
  const int n_cot = (a.c_out + 31) >> 5;
  const int h = lane >> 5;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // weight-fragment ring, RING k-steps deep: global k-step gk = 4 * stage + ks.  Loads are branch-free
  // (indices clamped to valid memory; a clamped fragment is either never used or feeds a tile that is
  // never stored) so that the compiler can count them with partial vmcnt waits.
  constexpr int RING = 4;
  s16x8 ring[RING][NT];
  const int gk_last = 4 * n_stage - 1;
  auto load_w = [&](int gk, s16x8 (&slot)[NT]) {
    gk = gk > gk_last ? gk_last : gk;
    const int sg = gk >> 2, ks = gk & 3;
    const bool main = sg < n_main;
    const unsigned short* wfr = main ? a.pw_w : a.res_w;
    const int kt = main ? a.kt_main : a.kt_res;
    const int kidx = (main ? sg : sg - n_main) * 4 + ks;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      int cot = cot0 + nt;
      cot = cot < n_cot ? cot : n_cot - 1;
      const u32x4 v = *reinterpret_cast<const u32x4*>(wfr + ((size_t)(cot * kt + kidx) * 64 + lane) * 8);
      slot[nt] = __builtin_bit_cast(s16x8, v);
    }
  };
#pragma unroll
  for (int r = 0; r < RING; ++r) load_w(r, ring[r]);

  {
    const int g = (lane >> 4) & 1;
    const int q4 = (lane >> 2) & 3;
    const int p4 = lane & 3;
    // per-lane LDS offsets of the transposed reads (stage-invariant)
    int aoff[KC / 16][MT][2];
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int c = 16 * ks + 8 * h + q4;
        const int t = 32 * mt + 16 * g + 4 * p4;
        aoff[ks][mt][0] = Tile::addr(c, t);
        aoff[ks][mt][1] = Tile::addr(c + 4, t);
      }
    for (int s = 0; s < n_stage; ++s) {
      stage_barrier();
      const char* src = dwt + (s & 1) * Tile::BYTES;
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks) {
        s16x8 af[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)src + aoff[ks][mt][0]));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)src + aoff[ks][mt][1]));
          af[mt] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], ring[ks][nt], acc[mt][nt], 0, 0, 0);
        load_w(4 * s + ks + RING, ring[ks]);
      }
    }
  }

  // =================================== epilogue =======================================================
  if constexpr (OUT_F32) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = (cot0 + nt) * 32 + (lane & 31);
      if (co < a.c_out) {
        const float bv = a.bias[co];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int t = t0 + 32 * mt + 8 * rg + 4 * h;
            float v0 = acc[mt][nt][4 * rg + 0] + bv, v1 = acc[mt][nt][4 * rg + 1] + bv;
            float v2 = acc[mt][nt][4 * rg + 2] + bv, v3 = acc[mt][nt][4 * rg + 3] + bv;
            if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            if (t < a.pitch_out)
              *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + (size_t)(b * a.c_out + co) * a.pitch_out + t) =
                  f32x4{v0, v1, v2, v3};
          }
        }
      }
    }
  } else {
    // bias + ReLU + bf16 pack, transposed through a wave-private LDS tile [NT*32 co][TT t] so that the
    // global stores are whole 16-B-per-lane row segments (the accumulator layout gives 8 B per lane
    // in 64 different rows per instruction, which is TA-issue-bound).
    constexpr int EP = TT * 2 + 16;                 // row pitch (bytes)
    char* const et = smem + (size_t)wave * (NT * 32) * EP;
    stage_barrier();                                // every consumer is done reading dwt (producers have exited)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      int co = (cot0 + nt) * 32 + (lane & 31);
      const float bv = a.bias[co < a.c_out ? co : 0];
      char* const row = et + (size_t)(nt * 32 + (lane & 31)) * EP;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          float v0 = acc[mt][nt][4 * rg + 0] + bv, v1 = acc[mt][nt][4 * rg + 1] + bv;
          float v2 = acc[mt][nt][4 * rg + 2] + bv, v3 = acc[mt][nt][4 * rg + 3] + bv;
          if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
          *reinterpret_cast<u32x2*>(row + (32 * mt + 8 * rg + 4 * h) * 2) = u32x2{pack_bf16(v0, v1), pack_bf16(v2, v3)};
        }
      }
    }
    // wave-private tile: LDS ops of one wave are processed in order, no barrier needed
    constexpr int LPR = TT / 8;                     // lanes per row (16 B each)
    constexpr int RPI = 64 / LPR;                   // rows per wave-instruction
    const int rsub = lane / LPR, csub = lane % LPR;
    unsigned short* const yb = reinterpret_cast<unsigned short*>(a.y);
#pragma unroll
    for (int r0 = 0; r0 < NT * 32; r0 += RPI) {
      const int rl = r0 + rsub;
      const int co = cot0 * 32 + rl;
      const int t = t0 + csub * 8;
      const u32x4 v = *reinterpret_cast<const u32x4*>(et + (size_t)rl * EP + csub * 16);
      if (co < a.c_out && t < a.pitch_out)
        *reinterpret_cast<u32x4*>(yb + (size_t)(b * a.c_out + co) * a.pitch_out + t) = v;
    }
  }
}

template <int TT, int NT, int STRIDE, bool DW, bool OUT_F32>
static int launch(const TcsArgs& a, int batch, hipStream_t stream) {
  constexpr int CO_WG = 4 * NT * 32;
  dim3 grid((a.t_out + TT - 1) / TT, batch, (round_up(a.c_out, 32) + CO_WG - 1) / CO_WG);
  size_t lds = (size_t)2 * KC * TT * 2;
  if (DW) lds += (size_t)KC * a.xpitch * 2 + (a.taps_lds ? (size_t)4 * NKMAX * 512 : 0);
  const size_t lds_epi = OUT_F32 ? 0 : (size_t)4 * NT * 32 * (TT * 2 + 16);
  if (lds < lds_epi) lds = lds_epi;
  auto kern = tcs_kernel<TT, NT, STRIDE, DW, OUT_F32>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, a);
  return hip_status(hipGetLastError());
}

}  // namespace ts

extern "C" int ts_time_pitch(int T) { return ts::round_up(T < 1 ? 1 : T, 128); }

extern "C" int ts_tcs_subblock_fwd(const ts_tcs_desc* d, const void* x, const int32_t* len_in, const void* x_res,
                                   const int32_t* len_res, void* y, void* stream_) {
  using namespace ts;
  if (!d || !x || !y || !len_in || !d->pw_w || !d->bias) return TS_EINVAL;
  if (d->batch <= 0 || d->c_in <= 0 || d->c_out <= 0 || d->t_out <= 0) return TS_EINVAL;
  if (d->pitch_in % 8 || d->pitch_out % 8 || d->pitch_out < d->t_out) return TS_EINVAL;
  if (d->stride < 1 || d->dilation < 1 || d->kernel < 1) return TS_EINVAL;
  if (d->stride > 1 && d->dilation > 1) return TS_EINVAL;          // blocks.py:192-193
  if (d->c_res > 0 && (!x_res || !len_res || !d->res_w || d->pitch_res % 8)) return TS_EINVAL;
  if (!d->depthwise && d->kernel != 1) return TS_EUNSUPPORTED;     // dense K>1 convs are not on the hot path
  if (d->depthwise && (!d->dw_taps || d->dw_ksteps <= 0 || d->dw_ksteps % NKP)) return TS_EINVAL;
  if (d->depthwise && d->stride > 2) return TS_EUNSUPPORTED;
  if (d->out_fp32 && d->depthwise) return TS_EUNSUPPORTED;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);

  TcsArgs a{};
  a.x = static_cast<const unsigned short*>(x);
  a.xres = static_cast<const unsigned short*>(x_res);
  a.y = y;
  a.len_in = len_in;
  a.len_res = len_res;
  a.taps = static_cast<const unsigned short*>(d->dw_taps);
  a.pw_w = static_cast<const unsigned short*>(d->pw_w);
  a.res_w = static_cast<const unsigned short*>(d->res_w);
  a.bias = d->bias;
  a.c_in = d->c_in; a.c_out = d->c_out; a.c_res = d->c_res;
  a.pitch_in = d->pitch_in; a.pitch_out = d->pitch_out; a.pitch_res = d->pitch_res;
  a.t_out = d->t_out;
  a.kernel = d->kernel; a.stride = d->stride; a.dilation = d->dilation; a.padding = d->padding;
  a.relu = d->relu;
  a.res_stride = d->res_stride < 1 ? 1 : d->res_stride;
  a.kt_main = round_up(d->c_in, KC) / 16;
  a.kt_res = round_up(d->c_res > 0 ? d->c_res : 1, KC) / 16;

  const bool wide = round_up(d->c_out, 32) > 256;   // 512-channel tiles for the wide layers
  const int TT = wide ? 64 : 128;
  if (d->depthwise) {
    a.npass = d->dw_ksteps / NKP;
    a.taps_lds = d->dw_ksteps <= NKMAX;
    const int padl4 = round_up(d->padding, 4);
    a.padl8 = round_up(padl4, 8);
    a.woff = a.padl8 - padl4;
    a.xe = round_up(TT * d->stride + 4 * d->dw_ksteps + 8, 8);
    if (a.xe > 64 * XMAX) return TS_EUNSUPPORTED;
    const int xpd = round_up(a.xe / 2 - 2, 16) + 2;   // dwords per row == 2 (mod 16): conflict-free window reads
    a.xpitch = xpd * 2;
    if (d->stride == 1)
      return wide ? launch<64, 4, 1, true, false>(a, d->batch, stream) : launch<128, 2, 1, true, false>(a, d->batch, stream);
    return wide ? launch<64, 4, 2, true, false>(a, d->batch, stream) : launch<128, 2, 2, true, false>(a, d->batch, stream);
  }
  // pointwise only: `stride` is handled by the staging (generic gather when > 1)
  if (d->out_fp32) {
    if (d->stride != 1) return TS_EUNSUPPORTED;
    return launch<128, 2, 1, false, true>(a, d->batch, stream);
  }
  if (d->stride == 1)
    return wide ? launch<64, 4, 1, false, false>(a, d->batch, stream) : launch<128, 2, 1, false, false>(a, d->batch, stream);
  if (d->stride == 2)
    return wide ? launch<64, 4, 2, false, false>(a, d->batch, stream) : launch<128, 2, 2, false, false>(a, d->batch, stream);
  return TS_EUNSUPPORTED;
}
