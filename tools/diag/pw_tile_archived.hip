// ARCHIVED EXPERIMENT (round 6), not part of the product build: measured equal to the staged kernel it was meant to replace -- see
// profiles/round6_c4_pointwise.md.  To rebuild it: copy to thunder_speech_amd/csrc/pw_tile.hip, declare launch_pw_tile in tcs_shared.hpp and call it
// from ts_tcs_subblock_fwd's masked pointwise-only branch.
// Whole-K pointwise tile kernel (gfx950): y[b, co, t] = act(sum_ci W[co, ci] * mask(x)[b, ci, t] + bias[co]) for the training path's 1x1
// convolutions -- the forward product of a MaskedConv1d(kernel_size=1) and its data gradient (W^T) in a train-mode QuartznetBlock /
// CitrinetBlock (reference quartznet/blocks.py:169-182 inside :317-338; autograd's conv backward for the data gradient).
//
// Why it exists.  At the fine-tuning batch (32 clips x 501 frames = 251 tiles of 64 frames on 256 compute units) every workgroup of the staged
// kernel (csrc/tcs_kernel.hip, pointwise-only mode) runs ONE tile: eight stages of 64 input channels, each a global -> register -> LDS copy
// requested one stage ahead, a barrier, 32 matrix instructions.  A stage cannot be shorter than a loaded HBM round trip (~2 us) while its
// arithmetic takes 0.43 us: 18.6 us per 512 x 512 launch, 18 % of the matrix peak, with 8 KB per compute unit in flight where ~32 KB are needed.
// Here the workgroup requests its WHOLE input tile (all c_in rows x 64 frames: 64 KB at 512 channels, 160 KB of LDS per CU hold it) before
// anything else, 128-channel chunk by chunk into registers; chunk c is written to its own LDS region (no buffer is ever reused, so ONE barrier
// per chunk), and all eight waves multiply chunk c while chunks c + 1 .. are still in flight.  Weights: the same pre-packed B fragments of
// v_mfma_f32_32x32x16_bf16 the staged kernel streams from L2 ([c_out / 32][c_in / 16][64 lanes][8]), through a 4-k-step register ring; A
// fragments: ds_read_b64_tr_b16 from the [ci][t] tile (DwTile<64> swizzle).  Epilogue = the staged kernel's: bias as the accumulators' initial
// value, optional ReLU on packed pairs, optional zeroed tail, per-tile BatchNorm statistics (ts_tcs_desc.stats), rows leave as 16-byte segments
// through a wave-private LDS tile (placed over the input tile once every wave is done with it).
#include "tcs_shared.hpp"

namespace ts {
namespace {

#ifndef TS_PWT_STORE
#define TS_PWT_STORE 0
#endif
#ifdef TS_PWT_STAMP
__device__ unsigned long long g_pwt_stamp[4096 * 8];
#define PWT_STAMP(i) if (lane == 0 && wave == 0 && blockIdx.x < 4096) g_pwt_stamp[blockIdx.x * 8 + (i)] = wall_clock64()
#else
#define PWT_STAMP(i)
#endif

template <int NCH, int NT, int RING>
__global__ __launch_bounds__(512) void pw_tile_kernel(const TcsArgs a) {
  constexpr int TT = 64, MT = TT / 32, EP = TT * 2 + 16;
  constexpr int KS_CH = 8, NKS = NCH * KS_CH;          // k-steps of 16 input channels: per 128-channel chunk, per tile
  constexpr int LPR = TT / 8, RPI = 64 / LPR;           // epilogue: lanes per 64-frame row (16 B each), rows per wave instruction
  using Tile = DwTile<TT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x;
  const int tt = tile % a.n_tt, z = (tile / a.n_tt) % a.n_z, b = (tile / a.n_tt) / a.n_z;
  const int t0 = tt * TT;
  PWT_STAMP(0);

  // ---- weight ring first (the first matrix instruction needs it together with chunk 0), then the whole input tile
  const int n_cot = (a.c_out + 31) >> 5;
  const int cot0 = (z * 8 + wave) * NT;                 // first 32-channel output tile of this wave
  const unsigned short* wptr[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int cotc = cot0 + nt < n_cot ? cot0 + nt : n_cot - 1;      // tiles beyond c_out: clamped, never stored
    wptr[nt] = a.pw_w + ((size_t)cotc * a.kt_main * 64 + lane) * 8;
  }
  s16x8 ring[RING][NT];
#pragma unroll
  for (int r = 0; r < RING; ++r)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ring[r][nt] = __builtin_bit_cast(s16x8, *reinterpret_cast<const u32x4*>(wptr[nt] + r * 512));

  const int row = tid >> 3, seg = tid & 7;              // thread = (row of a 64-row half chunk, 16-byte segment of its 128-byte tile row)
  const int tcol = t0 + seg * 8;
  const bool col_ok = tcol < a.pitch_in;
  const unsigned short* const src = a.x + ((size_t)b * a.c_in + row) * a.pitch_in + tcol;
  u32x4 X[2 * NCH];
#pragma unroll
  for (int i = 0; i < 2 * NCH; ++i)
    X[i] = col_ok ? *reinterpret_cast<const u32x4*>(src + (size_t)i * 64 * a.pitch_in) : u32x4{0u, 0u, 0u, 0u};
  const int nv = a.len_in[b] - tcol;                    // frames of this lane's group below the clip's length (MaskedConv1d input mask)

  f32x16 acc[MT][NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = (cot0 + j) * 32 + (lane & 31);
    const float bv = a.bias[col < a.c_out ? col : 0];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
  }
  const int h = lane >> 5, gq = (lane >> 4) & 1, q4 = (lane >> 2) & 3, p4 = lane & 3;
  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) abase[mt] = Tile::addr(8 * h + q4, 32 * mt + 16 * gq + 4 * p4);
  const int xw = Tile::addr(row, seg * 8);              // (rows row and row + 64 share the swizzle key: + 64 * ROWB)

  auto read_a = [&](const char* base, int ks, s16x8 (&af)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)base + abase[mt] + ks * 16 * Tile::ROWB));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)base + abase[mt] + ks * 16 * Tile::ROWB + 4 * Tile::ROWB));
      af[mt] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };

#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    char* const base = smem + (size_t)c * 128 * Tile::ROWB;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      u32x4 v = X[2 * c + rr];
      if (nv < 8) v = keep_first(v, nv);
      *reinterpret_cast<u32x4*>(base + rr * 64 * Tile::ROWB + xw) = v;
    }
    stage_barrier();                                    // chunk c is in LDS (its region is written once: no barrier behind the reads)
    if (c == 0) { PWT_STAMP(1); }
    if (c == NCH - 1) { PWT_STAMP(2); }
    s16x8 afA[MT], afB[MT];
    read_a(base, 0, afA);
#pragma unroll
    for (int ks = 0; ks < KS_CH; ++ks) {
      const int gk = c * KS_CH + ks;
      s16x8 (&cur)[MT] = (ks & 1) ? afB : afA;
      s16x8 (&nxt)[MT] = (ks & 1) ? afA : afB;
      if (ks + 1 < KS_CH) read_a(base, ks + 1, nxt);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[mt], ring[gk % RING][nt], acc[mt][nt], 0, 0, 0);
      if (gk + RING < NKS) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          ring[gk % RING][nt] = __builtin_bit_cast(s16x8, *reinterpret_cast<const u32x4*>(wptr[nt] + (size_t)(gk + RING) * 512));
      }
    }
  }

  // ---- epilogue (csrc/tcs_kernel.hip's, on wave-private LDS tiles laid over the input tile: every wave must be done reading it)
  stage_barrier();
  PWT_STAMP(3);
  char* const et = smem + (size_t)wave * 32 * EP;
  unsigned short* const yb = reinterpret_cast<unsigned short*>(a.y);
  const unsigned floor2 = a.relu ? 0u : 0x80008000u;
  const int len_out = a.zero_tail ? a.len_in[b] : 0x7fffffff;
  const bool partial = t0 + TT > len_out;
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");       // the accumulators settle before inline-asm readers
  if (a.stats) {
    // BatchNorm(train) statistics of this tile per output channel, out of the accumulators (frames < t_out; f32), channel-major
    const int nvf = a.t_out - t0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = 32 * mt + 8 * (r >> 2) + 4 * h + (r & 3) < nvf ? acc[mt][nt][r] : 0.f;
          s1 += v;
          s2 = fmaf(v, v, s2);
        }
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      const int co = (cot0 + nt) * 32 + (lane & 31);
      if (h == 0 && co < a.c_out) {
        float* const o = a.stats + ((size_t)co * (a.batch * a.n_tt) + (b * a.n_tt + tt)) * 2;
        o[0] = s1; o[1] = s2;
      }
    }
  }
  const int rsub = lane / LPR, csub = lane % LPR;
  const s16x2 f2 = __builtin_bit_cast(s16x2, floor2);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int cob = (cot0 + nt) * 32;
    char* const prow = et + (size_t)(lane & 31) * EP + 8 * h;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float v0 = acc[mt][nt][4 * rg + 0], v1 = acc[mt][nt][4 * rg + 1], v2 = acc[mt][nt][4 * rg + 2], v3 = acc[mt][nt][4 * rg + 3];
        if (partial) {
          const int t = t0 + 32 * mt + 8 * rg + 4 * h;
          v0 = t + 0 < len_out ? v0 : 0.f; v1 = t + 1 < len_out ? v1 : 0.f; v2 = t + 2 < len_out ? v2 : 0.f; v3 = t + 3 < len_out ? v3 : 0.f;
        }
        const unsigned lo = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16(v0, v1)), f2));
        const unsigned hi = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16(v2, v3)), f2));
        *reinterpret_cast<u32x2*>(prow + (32 * mt + 8 * rg) * 2) = u32x2{lo, hi};
      }
    }
    // wave-private tile: the LDS operations of one wave are processed in order, no barrier needed
#pragma unroll
    for (int r0 = 0; r0 < 32; r0 += RPI) {
      const int rl = r0 + rsub, co = cob + rl, t = t0 + csub * 8;
      const u32x4 v = *reinterpret_cast<const u32x4*>(et + (size_t)rl * EP + csub * 16);
      if (co < a.c_out && t < a.pitch_out) {
        u32x4* const dstp = reinterpret_cast<u32x4*>(yb + ((size_t)b * a.c_out + co) * a.pitch_out + t);
#if TS_PWT_STORE == 1
        // streaming stores: the rows start their way to memory while the kernel is still running, instead of sitting dirty in the XCD's L2
        // until the end-of-kernel write-back (which the next launch waits for): 15.8 -> 14.1 us at 512 x 512, 6.2 -> 5.6 at 256 x 256
        __builtin_nontemporal_store(v, dstp);
#else
        *dstp = v;
#endif
      }
    }
  }
  PWT_STAMP(4);
}

template <int NCH, int NT, int RING>
int launch(TcsArgs& a, hipStream_t stream) {
  constexpr int CO_WG = 8 * NT * 32;
  a.n_tt = (a.t_out + 63) / 64;
  a.n_z = (round_up(a.c_out, 32) + CO_WG - 1) / CO_WG;
  a.n_tiles = a.batch * a.n_tt * a.n_z;
  const size_t tile_b = (size_t)NCH * 128 * 128, epi_b = (size_t)8 * 32 * (64 * 2 + 16);
  const size_t lds = tile_b > epi_b ? tile_b : epi_b;
  auto kern = pw_tile_kernel<NCH, NT, RING>;
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TS_EINVAL;
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return TS_EUNSUPPORTED;
    attr_set[dev] = true;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL(kern, dim3(a.n_tiles), dim3(512), lds, stream, a);
  return hip_status(hipGetLastError());
}

int g_pw_tile = 1;

}  // namespace

int launch_pw_tile(TcsArgs& a, hipStream_t stream) {
  if (!g_pw_tile || a.c_res > 0 || a.stride != 1 || a.c_in % 128 || a.c_in < 256 || a.c_in > 1024) return TS_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(a.x) % 16 || reinterpret_cast<uintptr_t>(a.y) % 16) return TS_EUNSUPPORTED;
  const bool wide = round_up(a.c_out, 32) > 256;
  // (weight ring of 4 k-steps: 8, 12 and 16 measured the same +- 0.3 us -- the stream is bound by L2 bandwidth, not latency; profiles/round6_pw_tile.txt)
#define TS_PWT(NCH_) if (a.c_in == 128 * NCH_) return wide ? launch<NCH_, 2, 4>(a, stream) : launch<NCH_, 1, 4>(a, stream);
  TS_PWT(2) TS_PWT(4) TS_PWT(8)
#undef TS_PWT
  return TS_EUNSUPPORTED;
}

}  // namespace ts

#ifdef TS_PWT_STAMP
extern "C" int ts_debug_pwt_stamps(unsigned long long* dst) {
  return ts::hip_status(hipMemcpyFromSymbol(dst, HIP_SYMBOL(ts::g_pwt_stamp), sizeof(unsigned long long) * 4096 * 8));
}
#endif
extern "C" int ts_tcs_pointwise_select(int32_t mode) {
  const int old = ts::g_pw_tile;
  ts::g_pw_tile = mode ? 1 : 0;
  return old;
}
