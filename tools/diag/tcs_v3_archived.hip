// Fused time-channel-separable sub-block for gfx950 (MI355X), inference -- merged kernel (round 3).
//
//   y[b, co, t] = act( sum_ci Wf[co, ci] * dw[b, ci, t] + bias[co] + sum_cr Wr[co, cr] * xres[b, cr, t] )
//   dw[b, ci, t] = sum_u taps[ci, u] * x[b, ci, t + u - pad]                      (stride 1, dilation 1, tail-zero tensors)
//
// Same arithmetic and the same tensors as the split kernel of csrc/tcs_kernel.hip (reference: quartznet/blocks.py:166-182,
// :222, :332-337); what changes is who does the depthwise and on which instruction.
//
//  * EIGHT UNIFORM WAVES, no producer / consumer roles.  Measured on the split kernel (tools/diag/exp_split.py, TS_EXP): its
//    pointwise k-loop, its depthwise FIR and its memory skeleton add up instead of overlapping, because the depthwise sat in ONE
//    in-order wave per SIMD that issues ~260 instructions per stage while the eight consumer waves wait at the barrier.  Here
//    every wave owns a 96-frame x 64-channel piece of the 192 x 256 output tile AND eight of the 64 input channels of the
//    stage being produced, so the depthwise instruction stream is spread over all the waves of the workgroup.
//  * DEPTHWISE = TOEPLITZ x TIME SEGMENTS on v_mfma_f32_16x16x32_bf16.  For one channel, D[m][n] = y[16 n + m]: M = 16 output
//    frames of a segment, N = 16 segments (12 used: 192 frames), K = 32 input frames of chunk c,
//        A_c[m][k] = w[o + pad + 32 c + k - m]   (Toeplitz slice of the taps, 0 outside [0, K))
//        B_c[k][n] = x[t0 + 16 n + o + 32 c + k]
//    with o = -8 ceil(pad / 8) and NC = 1..3 chunks (K <= 81).  B_0 is ONE 16-byte global load per lane (lane (n, kg): 8
//    frames at 16 n + 8 kg), and B_c(n) = B_0(n + 2c): a DPP row shift by 2c lanes -- no LDS staging of the input rows at all.
//    A_c comes from a "sliding window" tap image in LDS (8 bytes = 4 taps per window start, plan.pack_dw_taps_t16): lane
//    (m, kg) reads windows 8 kg - m + 15 + 32 c and + 4, two aligned ds_read_b64, conflict-free.  3 MFMAs of 16 cycles per channel
//    and 192 frames, against 216 v_mfma_f32_4x4x4 (per 16 channels) before; checked stand-alone in tools/diag/probe_t16.hip.
//  * The depthwise result goes to the [ci][t] tile the pointwise MFMAs read with ds_read_b64_tr_b16 -- unchanged, as are the
//    weight fragments, the bias-initialised accumulators, the residual 1x1 conv as identity stages and the epilogue.
//  * One barrier per stage.  Iteration i consumes stage i out of dwt[i & 1] and produces stage i + 1 into the other buffer
//    (its rows were fetched during iteration i - 1, its tap image landed by DMA during iteration i - 1); rows, taps and weights
//    of later stages are requested in the middle of the iteration.  The stage stream runs on across tile boundaries.
#include "tcs_shared.hpp"

#include <cstdlib>

namespace ts {

namespace {

template <int SH>
__device__ __forceinline__ u32x4 row_shl(u32x4 v) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v[i], 0x100 + SH, 0xf, 0xf, true);
  return r;
}

// one stage of the flat (tile, stage) stream
struct Cursor {
  TilePos pos;
  int tile, s, valid;
};

}  // namespace

template <int NC>
__global__ __launch_bounds__(512) void tcs_v3_kernel(const TcsArgs a) {
  constexpr int MT = 3, NT = 2, WN = 4, FW = 96, TT = 192;
  constexpr int ROWB = 512, TILEB = KC * ROWB;
  constexpr int EP = FW * 2 + 24, ER = 16;
  constexpr int CH = (32 * NC + 16) * 8;          // bytes of a channel's tap image
  constexpr int WIMG = 8 * CH;                    // a wave's 8 channels
  constexpr int ND = WIMG / 1024;                 // DMA instructions per wave and stage
  static_assert(WIMG % 1024 == 0, "tap image of a wave must be whole KiB");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const dwt = smem;                                             // [2][KC][ROWB]
  char* const cons0 = smem + 2 * TILEB;                               // [8][ER][EP] epilogue tiles
  char* const timg0 = cons0 + 8 * ER * EP;                            // [8][WIMG] tap images
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_main = a.c_in / KC;
  const int n_res = a.c_res / KC;
  const int n_stage = n_main + n_res;
  int tile0 = blockIdx.x, tile_step = gridDim.x, tile_end = a.n_tiles;
  if (a.xcd) {
    const int per = (a.n_tiles + 7) >> 3, xcd = blockIdx.x & 7;
    tile0 = xcd * per + (blockIdx.x >> 3);
    tile_step = gridDim.x >> 3;
    tile_end = min(a.n_tiles, (xcd + 1) * per);
    if (tile0 >= tile_end) return;
  }
  auto taddr = [](int c, int t) { return c * ROWB + ((((t >> 3) ^ ((c & 3) * 5))) << 4) + ((t & 7) << 1); };
  constexpr int RSRC_FLAGS = 0x00020000;
  auto rsrc = [](const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, RSRC_FLAGS); };
  auto ld16 = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); };

  // ---------------------------------------------------------------- production side ----------------------------------------------
  const int n = lane & 15, kg = lane >> 4;        // depthwise: segment / Toeplitz row, 8-frame group
  char* const timg = timg0 + (size_t)wave * WIMG;
  const char* const tapr = timg + 8 * (8 * kg - n + 15);              // this lane's window start
  const __amdgpu_buffer_rsrc_t rx = rsrc(reinterpret_cast<const char*>(a.x) - TS_GUARD_BYTES);
  const __amdgpu_buffer_rsrc_t ri = rsrc(n_res ? a.xres : a.x);
  const i32x4 rt = raw_rsrc(a.taps_t16, (unsigned)n_main * (8 * WIMG));
  const int pitch2_in = a.pitch_in * 2, pitch2_res = a.pitch_res * 2;
  const int lane_x = (16 * n + 8 * kg) * 2;
  const int dw_out = taddr(8 * wave, 16 * n + 4 * kg);                // + c * ROWB: rows 8 wave + c keep (c & 3) only if ... see below
  // identity stages: the wave's 8 rows x 24 16-byte pieces = 3 pieces per lane
  int id_src[3], id_dst[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int p = j * 64 + lane, row = p / 24, col = p % 24;
    id_src[j] = (8 * wave + row) * pitch2_res + col * 16;
    id_dst[j] = taddr(8 * wave + row, col * 8);
  }
  u32x4 X[8];

  auto x_origin = [&](const Cursor& c) {
    return (c.pos.b * a.c_in * a.pitch_in + c.pos.tt * TT + a.t16_o) * 2 + TS_GUARD_BYTES + (c.s * KC + 8 * wave) * pitch2_in;
  };
  auto i_origin = [&](const Cursor& c) {
    return (c.pos.b * a.c_res * a.pitch_res + c.pos.tt * TT) * 2 + (c.s - n_main) * KC * pitch2_res;
  };
  auto advance = [&](Cursor& c) {
    if (++c.s == n_stage) {
      c.s = 0;
      c.tile += tile_step;
      c.valid = c.tile < tile_end;
      if (c.valid) c.pos.advance(a.n_tt, a.n_z);
    }
  };
  // rows of stage c -> X (depthwise stage: 8 channels; identity stage: 3 pieces)
  auto fetch_rows = [&](const Cursor& c) {
    if (!c.valid) return;
    if (c.s < n_main) {
      const int so = x_origin(c);
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) X[ch] = ld16(rx, lane_x, so + ch * pitch2_in);
    } else {
      const int so = i_origin(c);
#pragma unroll
      for (int j = 0; j < 3; ++j) X[j] = ld16(ri, id_src[j], so);
    }
  };
  // tap image of stage c -> LDS (this wave's 8 channels).  `after` pins the DMAs behind the MFMAs that consumed the old image.
  auto fetch_taps = [&](const Cursor& c, float after) {
    if (!c.valid || c.s >= n_main) return;
    const int so = (c.s * 8 + wave) * WIMG;
#pragma unroll
    for (int h = 0; h < ND; ++h) lds_dma16(rt, timg + h * 1024, lane * 16, so + h * 1024, after);
  };
  // stage c (rows in X, tap image in LDS) -> dw tile `dst`; channels [C0, C1)
  float pin = 0.f;
  auto produce_dw = [&](char* dst, auto c0, auto c1) {
    constexpr int C0 = decltype(c0)::value, C1 = decltype(c1)::value;
#pragma unroll
    for (int ch = C0; ch < C1; ++ch) {
      const u32x4 b0 = X[ch];
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        const char* tp = tapr + ch * CH + cc * 256;
        const u32x2 a0 = *reinterpret_cast<const u32x2*>(tp), a1 = *reinterpret_cast<const u32x2*>(tp + 32);
        const u32x4 av = {a0[0], a0[1], a1[0], a1[1]};
        const u32x4 bv = cc == 0 ? b0 : (cc == 1 ? row_shl<2>(b0) : row_shl<4>(b0));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16x8, av), __builtin_bit_cast(s16x8, bv), acc, 0, 0, 0);
      }
      // rows 8 wave + ch: the XOR swizzle of taddr depends on (row & 3) = (ch & 3) only (8 wave is a multiple of 4)
      *reinterpret_cast<u32x2*>(dst + (dw_out ^ (((ch & 3) * 5) << 4)) + ch * ROWB) = u32x2{pack_bf16(acc[0], acc[1]), pack_bf16(acc[2], acc[3])};
      pin = acc[0];
    }
  };
  auto produce_id = [&](char* dst) {
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<u32x4*>(dst + id_dst[j]) = X[j];
  };

  // ---------------------------------------------------------------- consumption side ---------------------------------------------
  char* const priv = cons0 + (size_t)wave * ER * EP;
  const __amdgpu_buffer_rsrc_t rwm = rsrc(a.pw_w);
  const __amdgpu_buffer_rsrc_t rwr = rsrc(n_res ? a.res_w : a.pw_w);
  const int wm = wave / WN, wn = wave % WN;
  const int n_cot = (a.c_out + 31) >> 5;
  const int h = lane >> 5;
  const int gq = (lane >> 4) & 1;
  const int q4 = (lane >> 2) & 3;
  const int p4 = lane & 3;
  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) abase[mt] = taddr(8 * h + q4, wm * FW + 32 * mt + 16 * gq + 4 * p4);
  const int rsub = lane >> 4, csub = lane & 15;
  const unsigned floor2 = a.relu ? 0u : 0x80008000u;
  const int lane_w = lane * 16;

  s16x8 W[4][NT];                                  // weight fragments of the stage being consumed (k-step, output tile)
  auto fetch_w = [&](const Cursor& c, auto k0, auto k1) {
    constexpr int K0 = decltype(k0)::value, K1 = decltype(k1)::value;
    if (!c.valid) return;
    const bool res = c.s >= n_main;
    const int kt = res ? a.kt_res : a.kt_main;
    const int s = res ? c.s - n_main : c.s;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int cot = (c.pos.z * WN + wn) * NT + nt;
      const int so = ((cot < n_cot ? cot : n_cot - 1) * kt + 4 * s) * 1024;
#pragma unroll
      for (int ks = K0; ks < K1; ++ks)
        W[ks][nt] = __builtin_bit_cast(s16x8, res ? ld16(rwr, lane_w + ks * 1024, so) : ld16(rwm, lane_w + ks * 1024, so));
    }
  };
  f32x16 acc[MT][NT];
  float bnext[NT];
  auto bias_fetch = [&](const TilePos& p) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = ((p.z * WN + wn) * NT + nt) * 32 + (lane & 31);
      bnext[nt] = a.bias[col < a.c_out ? col : 0];
    }
  };
  s16x8 af[MT];
  auto read_a = [&](const char* src, int ks) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)src + abase[mt] + ks * 16 * ROWB));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)src + abase[mt] + ks * 16 * ROWB + 4 * ROWB));
      af[mt] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };
  auto mfma_ks = [&](int ks) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], W[ks][nt], acc[mt][nt], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>; using I2 = std::integral_constant<int, 2>;
  using I4 = std::integral_constant<int, 4>; using I8 = std::integral_constant<int, 8>;

  // ---------------------------------------------------------------- prologue ------------------------------------------------------
  Cursor c2;                                       // the stage whose rows / taps are requested next
  c2.pos.init(tile0, tile_step, a.n_tt, a.n_z);
  c2.tile = tile0; c2.s = 0; c2.valid = 1;
  TilePos pos = c2.pos;                            // tile being consumed
  Cursor c1 = c2;                                  // the stage produced in the running iteration (consumed in the next)
  fetch_rows(c2);
  fetch_taps(c2, 0.f);
  fetch_w(c2, I0{}, I4{});
  bias_fetch(pos);
  vm_wait<0>();
  unsigned gs = 0;
  if (c2.s < n_main) produce_dw(dwt, I0{}, I8{}); else produce_id(dwt);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  advance(c2);
  fetch_taps(c2, pin);
  fetch_rows(c2);
  stage_barrier();                                 // stage 0 is in dwt[0]

  for (int tile = tile0; tile < tile_end; tile += tile_step) {
    const int b = pos.b, t0 = pos.tt * TT;
    const int cot0 = (pos.z * WN + wn) * NT;
    const int len_b = a.zero_tail ? a.len_in[b] : 0;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = bnext[j];
    for (int s = 0; s < n_stage; ++s, ++gs) {
      const char* const src = dwt + (gs & 1) * TILEB;
      char* const dst = dwt + ((gs + 1) & 1) * TILEB;
      c1 = c2;                                     // rows in X (requested an iteration ago), tap image requested an iteration ago
      advance(c2);
      // everything this wave requested in the previous iteration has had most of an iteration to arrive
      vm_wait<0>();
#ifdef TS_EXP
      const bool prod = c1.valid && !(a.exp & 2), prod_dw = prod && c1.s < n_main;
      const bool kl = !(a.exp & 4);
#define KL(x) if (kl) { x; }
#else
      const bool prod = c1.valid, prod_dw = c1.valid && c1.s < n_main;
#define KL(x) x;
#endif
      KL(read_a(src, 0);
      mfma_ks(0))
      if (prod_dw) produce_dw(dst, I0{}, I4{});
      KL(read_a(src, 1);
      mfma_ks(1))
      if (prod_dw) produce_dw(dst, I4{}, I8{}); else if (prod) produce_id(dst);
      // requests for later stages: tap image and rows of stage i + 2, weights of stage i + 1 (first half: slots 0, 1 are free)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#ifdef TS_EXP
      if (!(a.exp & 16)) {
#endif
      fetch_taps(c2, pin);
      fetch_w(c1, I0{}, I2{});
      fetch_rows(c2);
#ifdef TS_EXP
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      KL(read_a(src, 2);
      mfma_ks(2);
      read_a(src, 3);
      mfma_ks(3))
#ifdef TS_EXP
      if (!(a.exp & 16))
#endif
      fetch_w(c1, I2{}, I4{});
      stage_barrier();
    }
    // ---- epilogue ----------------------------------------------------------------------------------------------------------
    pos.advance_if(tile + tile_step < tile_end, a.n_tt, a.n_z);
    bias_fetch(pos);
    unsigned short* const yb = reinterpret_cast<unsigned short*>(a.y);
    int len_out = 0x7fffffff;
    if (a.zero_tail) len_out = conv_len(len_b, a.kernel, 1, a.padding, a.dilation);
    const int tw = t0 + wm * FW;
    const bool partial = tw + FW > len_out;
    u32x4 keep = u32x4{~0u, ~0u, ~0u, ~0u};
    if (partial) keep = keep_first(keep, len_out - (tw + csub * 8));
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#ifdef TS_EXP
    if (!(a.exp & 8))
#endif
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int cob = (cot0 + nt) * 32;
      const s16x2 f2 = __builtin_bit_cast(s16x2, floor2);
      u32x2 pk[MT * 4];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const unsigned lo = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2,
              pack_bf16_settled(acc[mt][nt][4 * rg + 0], acc[mt][nt][4 * rg + 1])), f2));
          const unsigned hi = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2,
              pack_bf16_settled(acc[mt][nt][4 * rg + 2], acc[mt][nt][4 * rg + 3])), f2));
          pk[mt * 4 + rg] = u32x2{lo, hi};
        }
#pragma unroll
      for (int half = 0; half < 32 / ER; ++half) {
        if (((lane & 31) / ER) == half) {
          char* const row = priv + (size_t)(lane & (ER - 1)) * EP + 8 * h;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) *reinterpret_cast<u32x2*>(row + (32 * mt + 8 * rg) * 2) = pk[mt * 4 + rg];
        }
        if (csub < FW / 8) {
          unsigned short* const yrow = yb + (size_t)(b * a.c_out + cob + half * ER + rsub) * a.pitch_out + tw + csub * 8;
          const char* const prow = priv + (size_t)rsub * EP + csub * 16;
          u32x4 v[ER / 4];
#pragma unroll
          for (int i = 0; i < ER / 4; ++i) {
            const u32x2* const pr = reinterpret_cast<const u32x2*>(prow + 4 * i * EP);
            v[i] = u32x4{pr[0][0], pr[0][1], pr[1][0], pr[1][1]};
          }
#pragma unroll
          for (int i = 0; i < ER / 4; ++i) {
            if (partial) v[i] &= keep;
#ifdef TS_EXP
            if (!(a.exp & 1))
#endif
            if (cob + half * ER + 4 * i + rsub < a.c_out) *reinterpret_cast<u32x4*>(yrow + (size_t)(4 * i) * a.pitch_out) = v[i];
          }
        }
      }
    }
  }
  vm_wait<0>();                                    // no DMA may still be heading for this workgroup's LDS when it is released
}

template <int NC>
static int launch_v3_nc(TcsArgs& a, hipStream_t stream) {
  constexpr int CH = (32 * NC + 16) * 8;
  const size_t lds = (size_t)2 * KC * 512 + (size_t)8 * 16 * (96 * 2 + 24) + (size_t)8 * 8 * CH;
  auto kern = tcs_v3_kernel<NC>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  const int n_cu = cu_count();
  const int grid = a.n_tiles < n_cu ? a.n_tiles : n_cu;
  static const bool no_xcd = getenv("TS_NO_XCD") != nullptr;
  a.xcd = (grid % 8 == 0 && !no_xcd) ? 1 : 0;
  (void)hipGetLastError();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, a);
  return hip_status(hipGetLastError());
}

int launch_v3(TcsArgs& a, hipStream_t stream) {
  constexpr int TT = 192, CO_WG = 256;
  if (!a.taps_t16 || a.t16_nc < 1 || a.t16_nc > 3) return TS_EUNSUPPORTED;
  if (a.c_in % KC || a.c_res % KC || a.c_in == 0) return TS_EUNSUPPORTED;
  a.n_tt = (a.t_out + TT - 1) / TT;
  a.n_z = (round_up(a.c_out, 32) + CO_WG - 1) / CO_WG;
  a.n_tiles = a.batch * a.n_tt * a.n_z;
  switch (a.t16_nc) {
    case 1: return launch_v3_nc<1>(a, stream);
    case 2: return launch_v3_nc<2>(a, stream);
    default: return launch_v3_nc<3>(a, stream);
  }
}

}  // namespace ts
