// Fused time-channel-separable sub-block for gfx950 (MI355X), inference.
//
//   y[b, co, t] = act( sum_ci Wf[co, ci] * mask(dw[b, ci, t]) + bias[co] + sum_cr Wr[co, cr] * mask(xres[b, cr, t*rs]) )
//   dw[b, ci, t] = sum_u taps[ci, u] * mask(x)[b, ci, t*stride + u*dil - pad]
//
// Replaces the reference's per-sub-block ATen chain (quartznet/blocks.py:166-182 masked_fill + conv1d
// (groups=C) + masked_fill + conv1d(k=1), :222 batch_norm, :332-337 residual add + relu).
//
// Two kernels live in this file (DESIGN.md section 3.1 has the measurements behind each step; the single-stream pipelined
// kernel that sat between them in round 1 is gone):
//   tcs_kernel        first design and generic fallback: 4 producer + 4 consumer waves, 64/128-frame tiles, masked
//                     producers for caller tensors; still runs the stride-2 stem, odd shapes and the fp32 decoder.
//   tcs_split_kernel  12 waves = 8 pointwise consumers + 4 depthwise producers, 96/192-frame tiles: the default for every
//                     depthwise / pointwise-only layer with tail-zero tensors.
// Common to all of them:
//  * layout NCT-p: bf16 [B][C][Tp], time contiguous.  A tile = TT output frames x CO_WG output channels of one
//    clip; the input channels are walked in stages of 64; PERSISTENT workgroups (one per CU) stride over the tiles.
//  * depthwise FIR on the matrix cores: v_mfma_f32_4x4x4_16b_bf16 computes 16 independent 4x4x4
//    products per instruction -- one block per channel.  For channel c the A block is a 4x4 slice of the
//    Toeplitz matrix of its taps (rows = 4 consecutive output frames; pre-shifted per row on the host),
//    the B block holds 4 consecutive input samples for each of 4 time runs.  A lane keeps its sliding
//    input window in registers.  4.1x the fp32-VALU FMA rate measured on MI355X (240 vs 58 TMAC/s).
//  * the depthwise result is stored [ci][t] (XOR-swizzled 16-B chunks) and consumed as the A operand of
//    v_mfma_f32_32x32x16_bf16 through ds_read_b64_tr_b16 (hardware transpose read); the B operand
//    (BN-folded pointwise weights) is pre-packed per lane on the host and streamed from L2 through a register ring.
//  * residual 1x1 conv = extra stages over the block input whose "depthwise" is a copy, accumulating into the same
//    registers; bias = initial accumulator value; ReLU + bf16 pack in the epilogue.
//  * a lone wave issues an instruction only every ~8 cycles on this chip, so the steady state is kept almost VALU-free.
#include "tcs_shared.hpp"

#ifndef TS_TCS_STORE_AUX
#define TS_TCS_STORE_AUX 0       // experiment: cache policy of the generic kernel's bf16 result stores (raw buffer aux: 16 = sc1); see profiles/round6_c4_pointwise.md
#endif


namespace ts {

constexpr int XMAX = 5;        // staged row length <= 64 * XMAX elements
constexpr int NKMAX = 24;      // taps are cached in LDS up to this many k-steps
constexpr int RING_BYTES = 8;   // weight-fragment prefetch depth: RING_BYTES KiB per wave in flight


template <int TT, int NT, int STRIDE, bool DW, bool OUT_F32, bool TLDS, bool TZ, int XJ, int NPASS>
__global__ __launch_bounds__(512, 2) void tcs_kernel(const TcsArgs a) {
  constexpr int MT = TT / 32;     // 32-frame MFMA row tiles
  constexpr int M = TT / 16;      // 4-frame steps per lane run (4 runs per channel)
  constexpr int RUN = TT / 4;
  constexpr int NPP = (M - 1) * STRIDE + NKP;   // window pairs live per pass
  constexpr int IDJ = TT / 64;    // identity staging: 16-B column groups per lane per row
  constexpr int EP = TT * 2 + 16; // epilogue tile row pitch (bytes)
  using Tile = DwTile<TT>;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const dwt = smem;                                   // [2][KC][TT] bf16 swizzled
  char* const epi = smem + 2 * Tile::BYTES;                 // [4 consumers][32][EP]
  char* const xs = epi + 4 * 32 * EP;                       // [4 producers][16][xpitch] bf16 (DW only)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int n_main = (a.c_in + KC - 1) / KC;
  const int n_res = a.c_res > 0 ? (a.c_res + KC - 1) / KC : 0;
  const int n_stage = n_main + n_res;
  const int tile_step = gridDim.x;

  if (wave >= 4) {
    // ================================= PRODUCER =======================================================
    const int pv = wave - 4;                     // producer index: channels [16 pv, 16 pv + 16) of the stage
    const int r8 = lane >> 3, sub = lane & 7;    // staging: rows r8, r8 + 8; 16-B column groups sub + 8j
    const int cw = pv * 16 + (lane >> 2);        // depthwise: channel inside the stage handled by this lane
    const int q = lane & 3;                      // depthwise: time run (B column) and Toeplitz row (A row)
    const int nk = a.npass * NKP;
    const int xj = a.xe >> 6;                    // 128-byte column groups per staged row
    char* const xs_w = xs + (size_t)pv * 16 * a.xpitch * 2;
    char* const tl_w = xs + (size_t)4 * 16 * a.xpitch * 2 + (size_t)pv * nk * 512;
    char* const xw0 = xs_w + ((size_t)r8 * a.xpitch + sub * 8) * 2;              // staging write address, row r8
    char* const xw1 = xw0 + (size_t)8 * a.xpitch * 2;                            // row r8 + 8
    const char* const xrow = xs_w + ((size_t)(cw - pv * 16) * a.xpitch + a.woff + q * RUN * STRIDE) * 2;
    const bool chan_full = (a.c_in % KC) == 0;

    u32x4 X[2][TZ ? (XJ > 0 ? XJ : 1) : XMAX];
    u32x2 T[TZ ? (NPASS > 0 ? NPASS * NKP : 1) : NKMAX];
    u32x4 I[2][IDJ];

    if constexpr (TZ) {
      // ---------------------------------------------------------------------------------------------------
      // Tail-zero input (every row is 0 from its length to the pitch, pitch has slack, buffer has guards):
      // no masks, no predicates, no bounds checks -- negative frames read the previous row's zero tail.
      // The stage stream is flat over (tile, source, chunk); loads run one stage ahead ACROSS tile boundaries.
      // ---------------------------------------------------------------------------------------------------
      // Two independent prefetch streams, each with ONE issue site inside the loops (a register set that is written
      // from several places costs a v_mov per register per stage to reconcile):
      //   DW stream: X/T registers <- (tile, chunk) of the next depthwise stage, possibly of the NEXT tile
      //   ID stream: I registers   <- next identity stage (residual chunk, or the main chunk of a pointwise-only layer)
      const size_t row8 = (size_t)8 * a.pitch_in;
      const size_t lane_x = (size_t)(pv * 16 + r8) * a.pitch_in + sub * 8;          // DW staging, element offset of row r8
      const size_t lane_t = ((size_t)pv * nk * 64 + lane) * 4;                       // tap fragments: [chunk][wave][k][lane][4]
      const size_t chunk_x = (size_t)KC * a.pitch_in;
      const size_t chunk_t = (size_t)KC * 4 * nk * 4;
      const int n_id = DW ? n_res : n_stage;                                          // identity stages per tile
      const size_t lane_i_main = (size_t)(pv * 16 + r8) * a.pitch_in + sub * 8;
      const size_t lane_i_res = (size_t)(pv * 16 + r8) * a.pitch_res + sub * 8;

      auto tile_origin = [&](int tile, int& b, int& t0) { b = (tile / a.n_tt) / a.n_z; t0 = (tile % a.n_tt) * TT; };

      // ---- DW stream state: next stage to issue
      int dw_tile = blockIdx.x, dw_chunk = 0;
      const unsigned short* dw_src = nullptr;            // element pointer of (dw_tile, dw_chunk), lane part excluded
      auto dw_seek = [&]() {                             // called when dw_tile changes
        if (dw_tile < a.n_tiles) {
          int b, t0; tile_origin(dw_tile, b, t0);
          dw_src = a.x + ((size_t)b * a.c_in * a.pitch_in + (t0 * STRIDE - a.padl8));
        }
      };
      auto dw_issue = [&]() {
        if (dw_tile >= a.n_tiles) return;
        const unsigned short* src = dw_src + lane_x;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int j = 0; j < XJ; ++j) X[rr][j] = *reinterpret_cast<const u32x4*>(src + rr * row8 + j * 64);
        if (++dw_chunk == n_main) { dw_chunk = 0; dw_tile += tile_step; dw_seek(); }
        else dw_src += chunk_x;
      };
      // ---- ID stream state
      int id_tile = blockIdx.x, id_s = 0;                // id_s: identity stage inside the tile
      int id_b = 0, id_t0 = 0;
      auto id_issue = [&]() {
        if (id_tile >= a.n_tiles || n_id == 0) return;
        if (id_s == 0) tile_origin(id_tile, id_b, id_t0);
        const bool main = !DW && id_s < n_main;
        const int chunk = main ? id_s : (DW ? id_s : id_s - n_main);
        const unsigned short* src = main
            ? a.x + ((size_t)(id_b * a.c_in + chunk * KC) * a.pitch_in + id_t0) + lane_i_main
            : a.xres + ((size_t)(id_b * a.c_res + chunk * KC) * a.pitch_res + id_t0) + lane_i_res;
        const size_t pitch8 = (size_t)8 * (main ? a.pitch_in : a.pitch_res);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int j = 0; j < IDJ; ++j) I[rr][j] = *reinterpret_cast<const u32x4*>(src + rr * pitch8 + j * 64);
        if (++id_s == n_id) { id_s = 0; id_tile += tile_step; }
      };
      auto write_id = [&](char* dst) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int j = 0; j < IDJ; ++j)
            *reinterpret_cast<u32x4*>(dst + Tile::addr(pv * 16 + r8 + 8 * rr, (sub + 8 * j) * 8)) = I[rr][j];
      };

      unsigned gs = 0;
      if constexpr (DW) { dw_seek(); dw_issue(); }
      id_issue();
      for (int tile = blockIdx.x; tile < a.n_tiles; tile += tile_step) {
        if constexpr (DW) {
          for (int s = 0; s < n_main; ++s, ++gs) {
            char* const dst = dwt + (gs & 1) * Tile::BYTES;
            // taps of THIS stage straight into registers (NPASS is compile-time, so they are statically indexed
            // MFMA operands): they arrive from L2 while the staging writes below run -- no LDS round trip
            const unsigned short* tp = a.taps + (size_t)s * chunk_t + lane_t;
            if constexpr (TLDS) {
#pragma unroll
              for (int k = 0; k < NPASS * NKP; ++k) T[k] = *reinterpret_cast<const u32x2*>(tp + k * 256);
            }
            // staged registers -> wave-private LDS rows
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
              char* const dstb = rr ? xw1 : xw0;
#pragma unroll
              for (int j = 0; j < XJ; ++j) {
                u32x2* d2 = reinterpret_cast<u32x2*>(dstb + j * 128);
                d2[0] = u32x2{X[rr][j][0], X[rr][j][1]};
                d2[1] = u32x2{X[rr][j][2], X[rr][j][3]};
              }
            }
            dw_issue();                                   // next depthwise stage (may belong to the next tile)
            // depthwise FIR, LDS reads of pass p+1 issued before the MFMAs of pass p
            f32x4 d[M];
#pragma unroll
            for (int m = 0; m < M; ++m) d[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            s16x4 A0[NKP], A1[NKP], P0[NPP], P1[NPP];
            auto load_pass = [&](int pass, s16x4 (&A)[NKP], s16x4 (&P)[NPP]) {
#pragma unroll
              for (int k = 0; k < NKP; ++k) {
                if constexpr (TLDS) A[k] = __builtin_bit_cast(s16x4, T[pass * NKP + k]);
                else A[k] = *reinterpret_cast<const s16x4*>(tp + (pass * NKP + k) * 256);
              }
#pragma unroll
              for (int u = 0; u < NPP; ++u) P[u] = *reinterpret_cast<const s16x4*>(xrow + (pass * NKP + u) * 8);
            };
            auto mfma_pass = [&](const s16x4 (&A)[NKP], const s16x4 (&P)[NPP]) {
#pragma unroll
              for (int k = 0; k < NKP; ++k)
#pragma unroll
                for (int m = 0; m < M; ++m)
                  d[m] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(A[k], P[m * STRIDE + k], d[m], 0, 0, 0);
            };
            load_pass(0, A0, P0);
#pragma unroll
            for (int pass = 0; pass < NPASS; pass += 2) {
              if (pass + 1 < NPASS) load_pass(pass + 1, A1, P1);
              mfma_pass(A0, P0);
              if (pass + 1 < NPASS) {
                if (pass + 2 < NPASS) load_pass(pass + 2, A0, P0);
                mfma_pass(A1, P1);
              }
            }
#pragma unroll
            for (int m = 0; m < M; ++m)
              *reinterpret_cast<u32x2*>(dst + Tile::addr(cw, q * RUN + 4 * m)) =
                  u32x2{pack_bf16(d[m][0], d[m][1]), pack_bf16(d[m][2], d[m][3])};
            stage_barrier();
          }
        }
        for (int s = 0; s < n_id; ++s, ++gs) {
          write_id(dwt + (gs & 1) * Tile::BYTES);
          id_issue();
          stage_barrier();
        }
      }
      return;
    }

    unsigned gs = 0;                             // global stage counter of this workgroup (selects the dwt buffer)
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += tile_step) {
      const int tt = tile % a.n_tt;
      const int b = (tile / a.n_tt) / a.n_z;
      const int t0 = tt * TT;
      const int len_in = a.len_in[b];
      const int len_res = a.c_res > 0 ? a.len_res[b] : 0;
      const int tin0 = t0 * STRIDE - a.padl8;
      const int len_mid = DW ? conv_len(len_in, a.kernel, STRIDE, a.padding, a.dilation) : 0;
      // interior tile: whole staged window inside the clip's valid frames -> no masks, no predicates
      const bool fast = DW && chan_full && tin0 >= 0 && tin0 + a.xuse <= len_in && tin0 + a.xe <= a.pitch_in &&
                        t0 + TT <= len_mid;

      // ---- identity staging (pointwise-only main source, residual source) ----------------------------
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
          const bool full = t0 + TT <= len;
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int cl = pv * 16 + r8 + 8 * rr;
#pragma unroll
            for (int j = 0; j < IDJ; ++j) {
              const int tl = (sub + 8 * j) * 8;
              *reinterpret_cast<u32x4*>(dst + Tile::addr(cl, tl)) = full ? I[rr][j] : keep_first(I[rr][j], len - (t0 + tl));
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
        // per-tile lane-invariant global row pointers (element units)
        const unsigned short* const xg0 = a.x + ((size_t)(b * a.c_in + pv * 16 + r8) * a.pitch_in + tin0 + sub * 8);
        const size_t row8 = (size_t)8 * a.pitch_in;
        const size_t chunk_stride = (size_t)KC * a.pitch_in;
        const unsigned short* const tg0 = a.taps + ((size_t)pv * nk * 64 + lane) * 4;   // [chunk][wave][k][lane][4]
        const size_t tap_chunk = (size_t)KC * 4 * nk * 4;

        // `fast` is wave-uniform: only the small predicate / mask pieces are duplicated, the depthwise body is shared
        auto issue_x = [&](int chunk) {
          const unsigned short* src0 = xg0 + chunk * chunk_stride;
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const unsigned short* src = src0 + rr * row8;
            const int c = chunk * KC + pv * 16 + r8 + 8 * rr;
#pragma unroll
            for (int j = 0; j < XMAX; ++j) {
              if (j < xj) {
                if (fast) {
                  X[rr][j] = *reinterpret_cast<const u32x4*>(src + j * 64);
                } else {
                  const int t = tin0 + 8 * (sub + 8 * j);
                  X[rr][j] = u32x4{0u, 0u, 0u, 0u};
                  if (c < a.c_in && t >= 0 && t < a.pitch_in) X[rr][j] = *reinterpret_cast<const u32x4*>(src + j * 64);
                }
              }
            }
          }
        };
        auto write_x = [&]() {
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            char* const dstb = rr ? xw1 : xw0;
#pragma unroll
            for (int j = 0; j < XMAX; ++j) {
              if (j < xj) {
                u32x4 v = X[rr][j];
                if (!fast) v = keep_first(v, len_in - (tin0 + 8 * (sub + 8 * j)));
                u32x2* dst = reinterpret_cast<u32x2*>(dstb + j * 128);
                dst[0] = u32x2{v[0], v[1]};
                dst[1] = u32x2{v[2], v[3]};
              }
            }
          }
        };
        auto issue_t = [&](int chunk) {
          if constexpr (!TLDS) return;
          const unsigned short* tp = tg0 + chunk * tap_chunk;
#pragma unroll
          for (int p = 0; p < NKMAX / NKP; ++p)
            if (p < a.npass) {
#pragma unroll
              for (int k = 0; k < NKP; ++k) T[p * NKP + k] = *reinterpret_cast<const u32x2*>(tp + (p * NKP + k) * 256);
            }
        };
        auto write_t = [&]() {
          if constexpr (!TLDS) return;
#pragma unroll
          for (int p = 0; p < NKMAX / NKP; ++p)
            if (p < a.npass) {
#pragma unroll
              for (int k = 0; k < NKP; ++k) *reinterpret_cast<u32x2*>(tl_w + ((p * NKP + k) * 64 + lane) * 8) = T[p * NKP + k];
            }
        };
        auto depthwise = [&](int chunk, char* dst) {
          f32x4 d[M];
#pragma unroll
          for (int m = 0; m < M; ++m) d[m] = f32x4{0.f, 0.f, 0.f, 0.f};
          const unsigned short* tp = tg0 + chunk * tap_chunk;
          for (int pass = 0; pass < a.npass; ++pass) {
            s16x4 A[NKP];
#pragma unroll
            for (int k = 0; k < NKP; ++k) {
              // TLDS is a template parameter on purpose: a runtime select between an LDS and a global pointer
              // compiles to flat_load + s_waitcnt vmcnt(0) lgkmcnt(0), which drains every prefetch in flight
              if constexpr (TLDS) A[k] = *reinterpret_cast<const s16x4*>(tl_w + ((pass * NKP + k) * 64 + lane) * 8);
              else A[k] = *reinterpret_cast<const s16x4*>(tp + (pass * NKP + k) * 256);
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
          // frames >= len_mid are zero for the pointwise conv (quirk A2: its input is re-masked)
#pragma unroll
          for (int m = 0; m < M; ++m) {
            const int tl = q * RUN + 4 * m;
            f32x4 v = d[m];
            if (!fast) {
#pragma unroll
              for (int i = 0; i < 4; ++i) v[i] = (t0 + tl + i < len_mid) ? v[i] : 0.f;
            }
            *reinterpret_cast<u32x2*>(dst + Tile::addr(cw, tl)) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
          }
        };
        issue_x(0);
        issue_t(0);
        for (int s = 0; s < n_main; ++s, ++gs) {
          write_x();
          write_t();
          if (s + 1 < n_main) { issue_x(s + 1); issue_t(s + 1); }
          else if (n_res > 0) issue_id(n_main);
          depthwise(s, dwt + (gs & 1) * Tile::BYTES);
          stage_barrier();
        }
      } else {
        issue_id(0);
      }
      // identity stages: all stages when !DW, else the residual ones (first one already prefetched)
      for (int s = DW ? n_main : 0; s < n_stage; ++s, ++gs) {
        write_id(s, dwt + (gs & 1) * Tile::BYTES);
        if (s + 1 < n_stage) issue_id(s + 1);
        stage_barrier();
      }
    }
    return;
  }

  // =================================== CONSUMER =======================================================
  const int n_cot = (a.c_out + 31) >> 5;
  const int h = lane >> 5;
  const int gq = (lane >> 4) & 1;
  const int q4 = (lane >> 2) & 3;
  const int p4 = lane & 3;
  // per-lane LDS offsets of the transposed A-operand reads.  The XOR swizzle depends only on (c & 3) and
  // ((c >> 1) & 1), which the +16 ks and +4 row steps leave unchanged, so one register per 32-frame tile is
  // enough; the k-step and the +4-row offsets are immediates.
  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) abase[mt] = Tile::addr(8 * h + q4, 32 * mt + 16 * gq + 4 * p4);
  char* const et = epi + (size_t)wave * 32 * EP;
  constexpr int LPR = TT / 8;                     // epilogue: lanes per output row (16 B each)
  constexpr int RPI = 64 / LPR;                   // rows per wave-instruction
  const int rsub = lane / LPR, csub = lane % LPR;

  unsigned gs = 0;
  for (int tile = blockIdx.x; tile < a.n_tiles; tile += tile_step) {
    const int tt = tile % a.n_tt;
    const int z = (tile / a.n_tt) % a.n_z;
    const int b = (tile / a.n_tt) / a.n_z;
    const int t0 = tt * TT;
    const int cot0 = (z * 4 + wave) * NT;          // first 32-channel output tile of this wave

    // the accumulators start at the (BN-folded) bias of the lane's output channel: no add in the epilogue
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

    // weight-fragment ring, RING k-steps deep.  One running pointer per 32-channel output tile walks the fragment
    // stream (1 KiB per k-step) of the main weights, switches to the residual weights after 4 * n_main k-steps and
    // stops advancing at the end (the surplus loads re-read the last fragment and are never used).  Tiles beyond
    // c_out are clamped to the last valid tile; their results are never stored.  Branch-free loads, so the compiler
    // counts them with partial vmcnt waits.
    constexpr int RING = RING_BYTES / NT;          // k-steps ahead (4 for NT = 2, 2 for NT = 4)
    s16x8 ring[RING][NT];
    const unsigned short* wptr[NT];
    int cotc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      cotc[nt] = cot0 + nt < n_cot ? cot0 + nt : n_cot - 1;
      wptr[nt] = a.pw_w + ((size_t)cotc[nt] * a.kt_main * 64 + lane) * 8;
    }
    int gk_next = 0;                               // k-step the pointers refer to
    const int gk_main = 4 * n_main, gk_end = 4 * n_stage;
    auto load_w = [&](s16x8 (&slot)[NT]) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        slot[nt] = __builtin_bit_cast(s16x8, *reinterpret_cast<const u32x4*>(wptr[nt]));
      ++gk_next;
      if (gk_next == gk_main && gk_main < gk_end) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wptr[nt] = a.res_w + ((size_t)cotc[nt] * a.kt_res * 64 + lane) * 8;
      } else if (gk_next < gk_end) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wptr[nt] += 64 * 8;
      }
    };
#pragma unroll
    for (int r = 0; r < RING; ++r) load_w(ring[r]);

    for (int s = 0; s < n_stage; ++s, ++gs) {
      stage_barrier();
      const char* src = dwt + (gs & 1) * Tile::BYTES;
      // A fragments of k-step ks+1 are read (transposed) from LDS before the MFMAs of k-step ks are issued
      s16x8 afA[MT], afB[MT];
      auto read_a = [&](int ks, s16x8 (&af)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (TS_LDS s16x4*)((TS_LDS char*)src + abase[mt] + ks * 16 * Tile::ROWB));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (TS_LDS s16x4*)((TS_LDS char*)src + abase[mt] + ks * 16 * Tile::ROWB + 4 * Tile::ROWB));
          af[mt] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      };
      auto mfma_ks = [&](int ks, const s16x8 (&af)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], ring[ks % RING][nt], acc[mt][nt], 0, 0, 0);
        load_w(ring[ks % RING]);
      };
      read_a(0, afA);
      read_a(1, afB);
      mfma_ks(0, afA);
      read_a(2, afA);
      mfma_ks(1, afB);
      read_a(3, afB);
      mfma_ks(2, afA);
      mfma_ks(3, afB);
    }

    // ---- epilogue (overlaps the producers' first stage of the next tile) -------------------------------
    if constexpr (OUT_F32) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = (cot0 + nt) * 32 + (lane & 31);
        if (co < a.c_out) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              const int t = t0 + 32 * mt + 8 * rg + 4 * h;
              float v0 = acc[mt][nt][4 * rg + 0], v1 = acc[mt][nt][4 * rg + 1];
              float v2 = acc[mt][nt][4 * rg + 2], v3 = acc[mt][nt][4 * rg + 3];
              if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
              if (t < a.pitch_out)
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + (size_t)(b * a.c_out + co) * a.pitch_out + t) =
                    f32x4{v0, v1, v2, v3};
            }
          }
        }
      }
    } else {
      // bias + bf16 pack + ReLU on packed pairs, transposed through a wave-private LDS tile [32 co][TT t] per
      // 32-channel output tile so that the global stores are whole 16-B-per-lane row segments (the accumulator
      // layout gives 8 B per lane in 64 different rows per instruction, which is TA-issue-bound).
      unsigned short* const yb = reinterpret_cast<unsigned short*>(a.y);
#if TS_TCS_STORE_AUX
      const __amdgpu_buffer_rsrc_t ry_out = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, 0x7fffffff, 0x00020000);
#endif
      // ReLU as a packed signed-16-bit max against `floor`: 0 clamps negative bf16 to +0, 0x8000 is a no-op
      const unsigned floor2 = a.relu ? 0u : 0x80008000u;
      // frames >= the output length are stored as 0 when the caller asks for the tail-zero invariant
      int len_out = 0x7fffffff;
      if (a.zero_tail) {
        len_out = a.len_in[b];
        if (DW) len_out = conv_len(len_out, a.kernel, STRIDE, a.padding, a.dilation);
        else if (STRIDE > 1) len_out = conv_len(len_out, 1, STRIDE, 0, 1);
      }
      const bool partial = t0 + TT > len_out;
      // the accumulators were last written by MFMAs: let them settle before inline-asm readers (no auto wait states)
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
      if constexpr (!DW) {
        if (a.stats) {
          // BatchNorm(train) statistics of THIS tile, per output channel, out of the accumulators (frames < t_out; f32): the training path's
          // next depthwise launch sums the tiles' pairs itself, so no pass over the stored tensor and no launch of its own is needed for them
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
              // channel-major: the consumer's wave owns a channel and reads its tiles as one contiguous run
              float* const o = a.stats + ((size_t)co * (a.batch * a.n_tt) + (b * a.n_tt + tt)) * 2;
              o[0] = s1; o[1] = s2;
            }
          }
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int cob = (cot0 + nt) * 32;
        char* const row = et + (size_t)(lane & 31) * EP + 8 * h;
        const s16x2 f2 = __builtin_bit_cast(s16x2, floor2);
        if (!partial) {                           // wave-uniform: the common tile has no frame beyond the length
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              const unsigned lo = __builtin_bit_cast(unsigned, __builtin_elementwise_max(
                  __builtin_bit_cast(s16x2, pack_bf16_settled(acc[mt][nt][4 * rg + 0], acc[mt][nt][4 * rg + 1])), f2));
              const unsigned hi = __builtin_bit_cast(unsigned, __builtin_elementwise_max(
                  __builtin_bit_cast(s16x2, pack_bf16_settled(acc[mt][nt][4 * rg + 2], acc[mt][nt][4 * rg + 3])), f2));
              *reinterpret_cast<u32x2*>(row + (32 * mt + 8 * rg) * 2) = u32x2{lo, hi};
            }
          }
        } else {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              const int t = t0 + 32 * mt + 8 * rg + 4 * h;
              const float v0 = t + 0 < len_out ? acc[mt][nt][4 * rg + 0] : 0.f, v1 = t + 1 < len_out ? acc[mt][nt][4 * rg + 1] : 0.f;
              const float v2 = t + 2 < len_out ? acc[mt][nt][4 * rg + 2] : 0.f, v3 = t + 3 < len_out ? acc[mt][nt][4 * rg + 3] : 0.f;
              const unsigned lo = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16_settled(v0, v1)), f2));
              const unsigned hi = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16_settled(v2, v3)), f2));
              *reinterpret_cast<u32x2*>(row + (32 * mt + 8 * rg) * 2) = u32x2{lo, hi};
            }
          }
        }
        // wave-private tile: LDS operations of one wave are processed in order, no barrier needed
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += RPI) {
          const int rl = r0 + rsub;
          const int co = cob + rl;
          const int t = t0 + csub * 8;
          const u32x4 v = *reinterpret_cast<const u32x4*>(et + (size_t)rl * EP + csub * 16);
          if (co < a.c_out && t < a.pitch_out) {
#if TS_TCS_STORE_AUX
            __builtin_amdgcn_raw_buffer_store_b128(v, ry_out, (int)(((size_t)(b * a.c_out + co) * a.pitch_out + t) * 2), 0, TS_TCS_STORE_AUX);
#else
            *reinterpret_cast<u32x4*>(yb + (size_t)(b * a.c_out + co) * a.pitch_out + t) = v;
#endif
          }
        }
      }
    }
  }
}

template <int TT, int NT, int STRIDE, bool DW, bool OUT_F32, bool TLDS = false, bool TZ = false, int XJ = 0, int NPASS = 0>
static int launch(TcsArgs& a, hipStream_t stream) {
  constexpr int CO_WG = 4 * NT * 32;
  a.n_tt = (a.t_out + TT - 1) / TT;
  a.n_z = (round_up(a.c_out, 32) + CO_WG - 1) / CO_WG;
  a.n_tiles = a.batch * a.n_tt * a.n_z;
  size_t lds = (size_t)2 * KC * TT * 2 + (size_t)4 * 32 * (TT * 2 + 16);
  if (DW) lds += (size_t)KC * a.xpitch * 2 + (a.taps_lds ? (size_t)4 * a.npass * NKP * 512 : 0);
  if (lds > 160 * 1024) return TS_EUNSUPPORTED;
  auto kern = tcs_kernel<TT, NT, STRIDE, DW, OUT_F32, TLDS, TZ, XJ, NPASS>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int n_cu = cu_count();
  const int grid = a.n_tiles < n_cu ? a.n_tiles : n_cu;      // persistent: one workgroup per CU
  (void)hipGetLastError();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, a);
  return hip_status(hipGetLastError());
}

}  // namespace ts

extern "C" int ts_time_pitch(int T) { return ts::round_up((T < 1 ? 1 : T) + 384, 128); }

/* frames per tile of the masked pointwise-only launch (depthwise = 0, kernel 1, stride 1, no TS_TCS_IN_TAILZERO) for this shape: the tile grid of
   ts_tcs_desc.stats is batch x ceil(t_out / this) */
extern "C" int ts_tcs_pointwise_tile_frames(int32_t batch, int32_t c_out, int32_t t_out) {
  using namespace ts;
  if (batch <= 0 || c_out <= 0 || t_out <= 0) return TS_EINVAL;
  if (round_up(c_out, 32) > 256) return 64;
  const int n_tt = (t_out + 127) / 128;
  return (long long)batch * n_tt * ((round_up(c_out, 32) + 255) / 256) < cu_count() ? 64 : 128;
}

static int g_pw_wide = 0;          // ts_tcs_pointwise_wide: 1 = wide-frame consumers for tail-zero pointwise-only layers with c_out > 256 (measured: no gain, profiles/round6_tcs_256.txt), 0 (default) = the 96 x 512 tiles

// one layer through the split kernel (csrc/tcs_split.hip)
static int split_single(const ts::TcsArgs& w, int npass, int xe, int wm, int dil, hipStream_t stream) {
  using namespace ts;
  // 32-bit byte offsets inside the split kernel's buffer descriptors
  const int64_t cmax = w.c_in > w.c_out ? w.c_in : w.c_out;
  if ((int64_t)w.batch * cmax * (w.pitch_in > w.pitch_out ? w.pitch_in : w.pitch_out) * 2 + TS_GUARD_BYTES >= (1ll << 31)) return TS_EUNSUPPORTED;
  if (w.c_res > 0 && (int64_t)w.batch * w.c_res * w.pitch_res * 2 >= (1ll << 31)) return TS_EUNSUPPORTED;
  SplitArgs a{};
  SplitLayer& L = a.layer;
  L.x = w.x; L.xres = w.xres; L.y = static_cast<unsigned short*>(w.y);
  if (!w.pw_w16 || (w.c_res > 0 && !w.res_w16)) return TS_EUNSUPPORTED;
  L.taps_raw = w.taps_raw; L.pw_w = w.pw_w16; L.res_w = w.res_w16; L.bias = w.bias;
  L.c_in = w.c_in; L.c_res = w.c_res; L.pitch_res = w.c_res > 0 ? w.pitch_res : w.pitch_in; L.relu = w.relu;
  L.kt_main = w.kt_main; L.kt_res = w.kt_res;
  L.se_y = w.se_y; L.se_gate = w.se_gate;
  a.len = w.len_in;
  a.batch = w.batch; a.c_out = w.c_out; a.pitch_in = w.pitch_in; a.pitch_out = w.pitch_out; a.t_out = w.t_out;
  a.kernel = w.kernel; a.padding = w.padding; a.dilation = w.dilation;
  a.woff = w.woff; a.padl8 = w.padl8; a.zero_tail = w.zero_tail;
  return launch_split_layer(a, npass, xe, wm, dil, stream);
}

extern "C" int ts_tcs_pointwise_wide(int32_t on) {
  const int old = g_pw_wide;
  g_pw_wide = on ? 1 : 0;
  return old;
}

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
  a.taps_raw = static_cast<const unsigned short*>(d->dw_taps_raw);
  a.pw_w = static_cast<const unsigned short*>(d->pw_w);
  a.res_w = static_cast<const unsigned short*>(d->res_w);
  a.pw_w16 = static_cast<const unsigned short*>(d->pw_w16);
  a.res_w16 = static_cast<const unsigned short*>(d->res_w16);
  a.bias = d->bias;
  a.se_y = static_cast<const unsigned short*>(d->se_y);
  a.se_gate = d->se_gate;
  a.stats = d->stats;
  // per-tile BatchNorm statistics come out of the generic pointwise-only kernel's epilogue only (what the training path launches)
  if (a.stats && (d->depthwise || d->out_fp32 || d->stride != 1 || d->c_res > 0 || (d->flags & TS_TCS_IN_TAILZERO))) return TS_EUNSUPPORTED;
  // the squeeze-excite tail lives in the split kernel's pointwise-only launch (tail-zero rows, stride 1, bf16 result, se_y at y's pitch);
  // every other configuration answers TS_EUNSUPPORTED and the caller runs ts_se_apply_fwd as a separate pass
  if (a.se_y && (!a.se_gate || d->depthwise || d->stride != 1 || d->out_fp32 || d->c_res > 0 || !(d->flags & TS_TCS_IN_TAILZERO) ||
                 !(d->flags & TS_TCS_OUT_ZERO_TAIL) || d->c_in % KC || reinterpret_cast<uintptr_t>(a.se_y) % 16))
    return TS_EUNSUPPORTED;
  a.batch = d->batch;
  a.c_in = d->c_in; a.c_out = d->c_out; a.c_res = d->c_res;
  a.pitch_in = d->pitch_in; a.pitch_out = d->pitch_out; a.pitch_res = d->pitch_res;
  a.t_out = d->t_out;
  a.kernel = d->kernel; a.stride = d->stride; a.dilation = d->dilation; a.padding = d->padding;
  a.relu = d->relu;
  a.res_stride = d->res_stride < 1 ? 1 : d->res_stride;
  a.kt_main = round_up(d->c_in, KC) / 16;
  a.kt_res = round_up(d->c_res > 0 ? d->c_res : 1, KC) / 16;

  a.zero_tail = (d->flags & TS_TCS_OUT_ZERO_TAIL) ? 1 : 0;
  const bool wide = round_up(d->c_out, 32) > 256;   // 512-channel tiles for the wide layers
  const int TT = wide ? 64 : 128;
  const int n_tt = (d->t_out + TT - 1) / TT;
  // tail-zero fast kernels: rows are 0 from their length to the pitch, the pitch has slack for the tile
  // overreach and the buffer has zero guards, so the producers need no mask, predicate or bounds check
  // (they also skip the re-masking of the depthwise output, which is only invisible when the output tail is zeroed)
  bool tz = (d->flags & TS_TCS_IN_TAILZERO) && (d->flags & TS_TCS_OUT_ZERO_TAIL) && !d->out_fp32 && d->c_in % KC == 0 &&
            (d->c_res == 0 || (d->c_res % KC == 0 && a.res_stride == 1 && d->pitch_res >= n_tt * TT));
  if (d->depthwise) {
    a.npass = d->dw_ksteps / NKP;
    if (d->flags & TS_TCS_TAPS_PHASE) {
      // dilation 2 as two interleaved dilation-1 sequences; dw_taps are packed for (K, stride 1, dilation 1, padding / 2)
      const bool ok = (d->flags & TS_TCS_IN_TAILZERO) && (d->flags & TS_TCS_OUT_ZERO_TAIL) && d->stride == 1 && d->dilation == 2 &&
                      d->padding % 2 == 0 && d->c_in % KC == 0 && d->c_res == 0 && round_up(d->c_out, 32) > 256;
      if (!ok) return TS_EUNSUPPORTED;
      TcsArgs w = a;
      w.padl8 = 2 * round_up(d->padding / 2, 4);      // frames staged before the tile: even, so staged parity == frame parity
      w.woff = 0;
      const int n_ttp = (d->t_out + 95) / 96;
      const bool fits = a.npass == 8 && 24 + 4 * (5 + d->dw_ksteps) <= 160 && (n_ttp - 1) * 96 - w.padl8 + 320 <= d->pitch_in &&
                        d->pitch_in - d->t_in >= w.padl8 && d->pitch_out >= n_ttp * 96;
      return (fits && d->dw_taps_raw) ? split_single(w, 8, 320, 1, 2, stream) : TS_EUNSUPPORTED;
    }
    a.taps_lds = d->dw_ksteps <= NKMAX;
    const int padl4 = round_up(d->padding, 4);
    a.padl8 = round_up(padl4, 8);
    a.woff = a.padl8 - padl4;
    const int M = TT / 16, RUN = TT / 4;
    a.xuse = a.woff + 3 * RUN * d->stride + 4 * ((M - 1) * d->stride + d->dw_ksteps);
    a.xe = round_up(a.xuse, 64);
    if (a.xe > 64 * XMAX) return TS_EUNSUPPORTED;
    a.xpitch = a.xe + 4;                              // row pitch == 8 (mod 16) bytes: conflict-free window reads
    const int xj = a.xe / 64;
    tz = tz && (n_tt - 1) * TT * d->stride - a.padl8 + a.xe <= d->pitch_in && d->pitch_in - d->t_in >= a.padl8;
    if (tz && d->stride == 1 && d->dilation == 1 && a.npass <= 7 && d->dw_taps_raw) {
      // split kernel: 96-frame granules, its own window geometry
      const int WM = split_tile_wm(d->c_out);
      const int TTp = 96 * WM;
      const int n_ttp = (d->t_out + TTp - 1) / TTp;
      const int xe = round_up(a.woff + TTp + 4 * d->dw_ksteps, 64);
      const bool fits = (n_ttp - 1) * TTp - a.padl8 + xe <= d->pitch_in && d->pitch_out >= n_ttp * TTp &&
                        (d->c_res == 0 || d->pitch_res >= (n_ttp - 1) * TTp + round_up(TTp, 64));
      if (fits) {
        const int st = split_single(a, a.npass, xe, WM, 1, stream);
        if (st != TS_EUNSUPPORTED) return st;
      }
    }
    if (tz) {
      // straight-line instantiations (staged row groups XJ and depthwise passes NPASS are compile-time) for the
      // geometries of the reference models; anything else takes the generic kernel below
#define TS_TZ(TT_, NT_, S_, TL_, XJ_, NP_) \
      if (TT == TT_ && d->stride == S_ && a.taps_lds == TL_ && xj == XJ_ && a.npass == NP_) \
        return launch<TT_, NT_, S_, true, false, TL_, true, XJ_, NP_>(a, stream);
      TS_TZ(128, 2, 1, true, 3, 3) TS_TZ(128, 2, 1, true, 3, 4) TS_TZ(128, 2, 1, true, 3, 2) TS_TZ(128, 2, 1, true, 3, 1)
      TS_TZ(64, 4, 1, true, 2, 5) TS_TZ(64, 4, 1, true, 3, 6) TS_TZ(64, 4, 1, true, 3, 7) TS_TZ(64, 4, 1, true, 2, 3)
      TS_TZ(64, 4, 1, false, 4, 15)
      TS_TZ(128, 2, 2, true, 5, 4)
#undef TS_TZ
    }
    if (a.taps_lds) {
      if (d->stride == 1)
        return wide ? launch<64, 4, 1, true, false, true>(a, stream) : launch<128, 2, 1, true, false, true>(a, stream);
      return wide ? launch<64, 4, 2, true, false, true>(a, stream) : launch<128, 2, 2, true, false, true>(a, stream);
    }
    if (d->stride == 1)
      return wide ? launch<64, 4, 1, true, false, false>(a, stream) : launch<128, 2, 1, true, false, false>(a, stream);
    return wide ? launch<64, 4, 2, true, false, false>(a, stream) : launch<128, 2, 2, true, false, false>(a, stream);
  }
  // pointwise only: `stride` is handled by the staging (generic gather when > 1)
  if (d->out_fp32) {
    if (d->stride != 1) return TS_EUNSUPPORTED;
    const int st = launch_pw_logits(a, stream);          // the decoders: few output channels, a pure read stream (csrc/pw_logits.hip)
    if (st != TS_EUNSUPPORTED) return st;
    return launch<128, 2, 1, false, true>(a, stream);
  }
  if (d->stride == 1) {
    // without a depthwise stage a tail-zero input needs no mask whether or not the output tail is zeroed: frames >= length
    // come out as relu(shift), which is what the reference computes from its masked input (quirk A2)
    const bool tz_in = (d->flags & TS_TCS_IN_TAILZERO) && d->c_in % KC == 0;
    if (tz_in && d->c_res == 0) {
      // pointwise only: the split kernel with identity stages only (the layer's input plays the residual input's role)
      TcsArgs w = a;
      const int WM = round_up(d->c_out, 32) <= 256 ? 2 : 1;
      w.c_res = d->c_in; w.c_in = 0; w.xres = a.x; w.res_w = a.pw_w; w.res_w16 = a.pw_w16; w.kt_res = a.kt_main; w.pitch_res = d->pitch_in;
      w.len_res = a.len_in; w.woff = 0; w.padl8 = 0;
      if (g_pw_wide && WM == 1) {
        // wide-frame consumers for the layers of more than 256 output channels (see launch_split_layer, wm code 4)
        const int n_ttw = (d->t_out + 191) / 192;
        if (d->pitch_in >= n_ttw * 192 && d->pitch_out >= n_ttw * 192) {
          const int st = split_single(w, 2, 128, 4, 1, stream);
          if (st != TS_EUNSUPPORTED) return st;
        }
      }
      const int TTp = 96 * WM;
      const int n_ttp = (d->t_out + TTp - 1) / TTp;
      if (d->pitch_in >= (n_ttp - 1) * TTp + round_up(TTp, 64) && d->pitch_out >= n_ttp * TTp) {
        const int st = WM == 2 ? split_single(w, 2, 256, 2, 1, stream) : split_single(w, 2, 128, 1, 1, stream);
        if (st != TS_EUNSUPPORTED) return st;
      }
    }
    if (a.se_y) return TS_EUNSUPPORTED;
    if (tz && d->pitch_in >= n_tt * TT)
      return wide ? launch<64, 4, 1, false, false, false, true, 0>(a, stream) : launch<128, 2, 1, false, false, false, true, 0>(a, stream);
    // narrow layers whose 128-frame tiling leaves compute units idle (the training path's 32 clips x 501 frames: 128 tiles on 256 CUs) take
    // 64-frame tiles: 9.2 -> 7.1 us at 256 -> 256 channels, 12.9 -> 9.5 at 512 -> 256 (tools/diag/pw_tile_bench.py); for the wide layers the
    // same halving (64 x 256 tiles, two workgroups per CU) measured slower, 16.7 vs 15.4 us
    if (!wide && (long long)d->batch * n_tt * ((round_up(d->c_out, 32) + 255) / 256) < cu_count()) return launch<64, 2, 1, false, false>(a, stream);
    return wide ? launch<64, 4, 1, false, false>(a, stream) : launch<128, 2, 1, false, false>(a, stream);
  }
  if (d->stride == 2)
    return wide ? launch<64, 4, 2, false, false>(a, stream) : launch<128, 2, 2, false, false>(a, stream);
  return TS_EUNSUPPORTED;
}
