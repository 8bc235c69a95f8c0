// Split kernel of the fused time-channel-separable sub-block (gfx950): 12 waves = 8 pointwise consumers + 4 depthwise
// producers, 96 / 192-frame tiles, persistent workgroups, one sub-block per launch.
//
//   y_l[b, co, t] = act( sum_ci Wf_l[co, ci] * dw_l[b, ci, t] + bias_l[co] + sum_cr Wr_l[co, cr] * xres_l[b, cr, t] )
//   dw_l[b, ci, t] = sum_u taps_l[ci, u] * x_l[b, ci, t + u - pad],   x_l = y_(l-1) for l >= 1
//
// Replaces the reference's per-sub-block ATen chain (quartznet/blocks.py:166-182 masked_fill + conv1d(groups=C) + masked_fill +
// conv1d(k=1), :222 batch_norm, :332-337 residual add + relu).  DESIGN.md section 3.1 has the measurements behind each step.
//
// Roles (every SIMD holds two consumer waves and one producer wave, 168 VGPRs each):
//   producer p (wave 8 + p): channels [16p, 16p + 16) of a 64-channel stage.  Rows global -> registers (two stages ahead) ->
//     wave-private LDS rows; raw tap image global -> LDS by DMA a stage ahead; depthwise FIR on v_mfma_f32_4x4x4_16b_bf16
//     (16 independent 4x4x4 products = 16 channels; A = Toeplitz slice of the taps, B = 4 consecutive samples of 4 time runs; the
//     lane's input window slides through registers); result -> dw tile [ci][t], XOR-swizzled 16-byte chunks.
//   consumer w: 96 frames x 64 output channels, v_mfma_f32_32x32x16_bf16; A = the dw tile read with ds_read_b64_tr_b16, B =
//     BN-folded weight fragments streamed from L2 through a register ring; epilogue: bias (initial accumulator), ReLU on packed
//     bf16 pairs, transposed through a wave-private LDS tile so that the global stores are 16-byte row segments.
//   iteration i: producers write stage i into dwt[i & 1], consumers read stage i-1; ONE s_barrier per stage; the stream runs on
//   across tile boundaries, the epilogue of a tile sits behind the barrier that ends it, so the producers work through it.
//
// Round 4 also ran all repeats of a block as ONE persistent launch of this kernel (a chain: per-tile arrival counters, write-through stores,
// L1-bypassing loads).  It was bit-identical and removed the launch gaps, but lost more than that to the sc1 traffic and to register pressure
// (2.98-3.00 vs 2.87 ms per encoder pass, profiles/round4_chain_ab.txt, round4_tcs_same_box.txt) and never shipped switched on; round 5 took it out
// of the product (git history: commit 4a86010 has the full kernel, ts_tcs_chain_fwd and its tests).
#include "tcs_shared.hpp"

#ifndef TS_SPLIT_SWITCH_OFF
#define TS_SPLIT_SWITCH_OFF 0
#endif
#ifndef TS_SPLIT_DIRECT_STORE
#define TS_SPLIT_DIRECT_STORE 0    // 1: result stores straight from the accumulators (8 bytes per lane), no LDS transpose (experiment)
#endif
#ifndef TS_SPLIT_RING
#define TS_SPLIT_RING 2            // consumer weight fragments: k-steps in flight (see the consumer's `ring`; 1 = the round 4-5 loop, kept for A/B builds)
#endif
#ifndef TS_SPLIT_SWITCH_WM
#define TS_SPLIT_SWITCH_WM 2         // which tiling the switch-off applies to: 2 = 192 x 256 (c_out <= 256), 1 = 96 x 512
#endif
#ifndef TS_SPLIT_STORE_AUX
// cache policy of the result stores (raw buffer aux: 1 = sc0, 2 = nt, 16 = sc1).  sc1 = write-through at system scope: the rows do not wait dirty in the
// XCD's L2 for the end-of-kernel write-back; same-box A/B of the C2 encoder, 4 interleaved runs each: 2.812 2.792 2.801 2.801 ms (plain) vs 2.788 2.783
// 2.791 2.789 (sc1); nt (streaming) costs +3 % -- it also evicts what the next launch would hit (profiles/round6_c4_pointwise.md section 3).
// Layers of up to 512 output channels only: on Citrinet-1024 (C3) plain stores measured 0.6 % faster (8.64 vs 8.69 ms, twice).
#define TS_SPLIT_STORE_AUX 16
#endif

namespace ts {

// DIL == 2 (dilation-2 layers, K87 of QuartzNet): the even and the odd frames of a row are two independent dilation-1
// sequences (y[2s+p] = sum_u w[u] x[2(s+u)+p - pad], pad even).  The producers stage each row as [even | odd] halves,
// lane runs 0,1 filter the even half and 2,3 the odd half with dilation-1 tap fragments (no zero-stuffed Toeplitz rows:
// 24 k-steps instead of 45), and lane pairs re-interleave their results on the way into the dw tile.
// NT: 32-channel output tiles per consumer wave -- 2: a workgroup covers 512 (WM = 1) or 256 (WM = 2) output channels; 1 (WM = 1 only): 96 frames x
// 256 channels, for layers of at most 256 output channels whose 192-frame tiling would leave a compute unit a single tile per layer (nothing
// to overlap its prologue and epilogue with).
// SE: the epilogue closes a CitrinetBlock -- y = relu(gate[b][co] * se_y[b][co][t] + result) with the main branch's output se_y read as 16-byte row
// segments right where the result rows leave (what ts_se_apply_fwd did in a separate pass over three tensors).
template <int NPASS, int XJ, int MT, int WM, int DIL = 1, int NT = 2, bool SE = false>
__global__ __launch_bounds__(768) void tcs_split_kernel(const SplitArgs a) {
  constexpr int WN = 8 / WM;
  constexpr int FW = 32 * MT;
  constexpr int TT = FW * WM;
  constexpr int M = TT / 16, RUN = TT / 4;        // producer: 4 runs of RUN frames per channel, M steps of 4 frames
  constexpr int NK = NPASS * NKP;
  constexpr int NP = NK + M - 1;
  constexpr int ROWB = TT <= 128 ? 256 : 512;
  constexpr int TILEB = KC * ROWB;
  constexpr int EP = FW * 2 + 24;
  constexpr int XP = 2 * XJ;                      // 16 rows x 128*XJ bytes per wave, 1 KiB per instruction
  constexpr int IDP = (2 * TT + 63) / 64;         // identity rows: 16 rows x 2*TT bytes
  // Depthwise taps: RAW, not as Toeplitz fragments.  Per channel two copies of the zero-padded tap array wp[n] = w[n - 3 - d]
  // (copy 1 shifted by one element), their dwords interleaved (dword j of copy c at byte 8 j + 4 c), CST bytes per channel;
  // Toeplitz row i of k-step k is wp[4k + 3 - i .. +3] = dwords 2k + (i < 2) and the next one of copy (i even): one
  // ds_read2_b32.  CST = 16 (mod 32) bytes puts the 8 channels x 4 rows of a half-wave on 32 different banks.  Half the bytes of
  // the pre-shifted fragments, so the taps of a 16-channel group fit TWICE: the image of the NEXT stage is fetched by DMA at
  // the start of the running one (a whole stage ahead of its first use) and no DMA sits inside the MFMA passes.
  constexpr int CST = (16 * NK + 16) % 32 == 16 ? 16 * NK + 16 : 16 * NK + 32;
  constexpr int NTD = (16 * CST + 1023) / 1024;   // KiB (= DMA instructions) per 16-channel group and stage
  constexpr int TAPB = NTD * 1024;
  constexpr int XSB = 16 * (64 * XJ + 4) * 2;     // bytes of a producer's staged rows
  constexpr int ER = (WM == 2 || DIL == 2 || MT > 3) ? 16 : 32;   // output-channel rows of a consumer's epilogue tile
  constexpr int PHW = 32 * XJ;                    // DIL == 2: frames of a staged half row

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const dwt = smem;                                             // [2][KC][ROWB]
  char* const cons0 = smem + 2 * TILEB;                               // [8][ER][EP] epilogue tiles
  char* const prod0 = cons0 + 8 * ER * EP;                            // [4][XSB + 2 * TAPB]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware tile order: workgroup i runs on XCD i % 8, and each XCD has its own L2.  Give every XCD one
  // contiguous range of tiles so that the input halos neighbouring time tiles share are fetched into ONE L2.
  int tile0 = blockIdx.x, tile_step = gridDim.x, tile_end = a.n_tiles;
  if (a.xcd) {
    const int per = (a.n_tiles + 7) >> 3, xcd = blockIdx.x & 7;
    tile0 = xcd * per + (blockIdx.x >> 3);
    tile_step = gridDim.x >> 3;
    tile_end = min(a.n_tiles, (xcd + 1) * per);
    if (tile0 >= tile_end) return;
  }
  // dw tile [ci][t]: rows of ROWB bytes (a whole number of 256-byte bank rows, so every row starts on bank 0) with the 16-byte chunk index
  // XOR-ed by a per-row key.  The consumers' transposed reads take 8 rows per half-wave -- channels 8 kg + q (+ 4), kg = 0, 1 -- and two 8-byte
  // halves of 2 chunks each per row: the key's bits 1.. must differ for all eight rows.  (c & 3) * 5 alone repeats for rows c and c + 8: a
  // 2-way conflict on EVERY A-fragment read, which the round-5 SQ counter pass showed (SQ_LDS_BANK_CONFLICT = 44 % of SQ_LDS_IDX_ACTIVE at K63;
  // profiles/round5_tcs_sq_counters.md); bit 3 of the row folded into bit 1 of the key makes the eight rows' keys 0, 5, 10, 15, 2, 7, 8, 13.
  auto taddr = [](int c, int t) { return c * ROWB + ((((t >> 3) ^ ((c & 3) * 5) ^ (((c >> 3) & 1) << 1))) << 4) + ((t & 7) << 1); };
  constexpr int RSRC_FLAGS = 0x00020000;
  auto rsrc = [](const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, RSRC_FLAGS); };
  auto ld16 = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); };
  const SplitLayer& L = a.layer;
  unsigned gs = 0;

  if (wave >= 8) {
    // ================================= PRODUCER =======================================================
    const int pw = wave - 8;
    char* const xs = prod0 + (size_t)pw * (XSB + 2 * TAPB);
    char* const tapl = xs + XSB;                   // two tap images: stage ds reads tapl[ds & 1]
    const int xpitch = 64 * XJ + 4;
    const int row = lane >> 2, sub = lane & 3;     // staging: row of the wave's 16 channels, 16-byte group sub + 4j
    const int q = lane & 3;                        // depthwise: channel = row, time run q
    char* const xw = xs + ((size_t)row * xpitch + sub * (DIL == 2 ? 4 : 8)) * 2;
    const char* const xrow = DIL == 2 ? xs + ((size_t)row * xpitch + (q >> 1) * PHW + a.woff + (q & 1) * RUN) * 2
                                      : xs + ((size_t)row * xpitch + a.woff + q * RUN) * 2;
    // this lane's Toeplitz row inside a tap image: channel `row`, copy by the parity of q, one dword in for q < 2
    const int tap_off = row * CST + ((lane & 1) ? 0 : 4) + ((lane & 3) < 2 ? 8 : 0);
    // dilation 1: steps 2h and 2h + 1 of a run write the two halves of ONE 16-byte chunk (RUN is a multiple of 8 frames), so M / 2 addresses and an
    // immediate offset do -- three registers fewer in a wave that has none to spare (the K 15..75 instantiations spilled eight without them)
    static_assert(DIL == 2 || (RUN % 8 == 0 && M % 2 == 0), "chunk pairs of the depthwise result");
    constexpr int NDO = DIL == 2 ? M : M / 2;
    int dw_out[NDO];
#pragma unroll
    for (int m = 0; m < NDO; ++m)
      dw_out[m] = DIL == 2 ? taddr(pw * 16 + row, (q & 1) * 2 * RUN + 8 * m + 4 * (q >> 1)) : taddr(pw * 16 + row, q * RUN + 8 * m);
    const int lane_x = ((pw * 16 + row) * a.pitch_in + sub * 8) * 2;
    const int lane_t = pw * TAPB + lane * 16;      // [chunk][16-ch group][TAPB]
    const int id_out = taddr(pw * 16 + row, sub * 8);
    const int chunk_x = KC * a.pitch_in * 2;
    const int chunk_t = 4 * TAPB;

    // ROWS2 (96-frame tiles, dilation 1, an even number of 64-channel stages): a stage's rows are requested TWO stages before it
    // runs, into two register sets that alternate statically (the stage loop runs in pairs), and the stage start waits with a
    // counted vmcnt instead of draining the queue -- the rows of the stage in between stay in flight (a loaded HBM round trip is
    // longer than one stage).
    constexpr bool ROWS2 = WM == 1 && DIL == 1;
    u32x4 X[ROWS2 ? 2 : 1][XP];
    // identity rows in flight: one stage ahead; TWO (a second register set, alternating statically) when the residual has an even number of
    // stages -- an identity stage is as short as the consumers' k-loop, shorter than a loaded HBM round trip.
    constexpr bool ID2 = WM == 1 && MT <= 3;       // (the 192-frame identity rows are 6 registers per set: one set)
    u32x4 I[IDP], I2[ID2 ? IDP : 1];
    s16x4 P[NP];
    u32x2 T[NK];
    f32x4 d[M];
    constexpr int WD = 1 < NPASS ? 1 : NPASS;      // passes whose window / tap reads run ahead of the MFMAs
    const char* trow = tapl + tap_off;             // re-pointed at the running stage's image at every stage start
    auto xs_write = [&](const u32x4 (&XS)[XP]) {
#pragma unroll
      for (int j = 0; j < XP; ++j) {
        if constexpr (DIL == 2) {     // 8 frames -> 4 even + 4 odd (the staged span starts on an even frame)
          *reinterpret_cast<u32x2*>(xw + j * 32) = u32x2{__builtin_amdgcn_perm(XS[j][1], XS[j][0], 0x05040100u),
                                                         __builtin_amdgcn_perm(XS[j][3], XS[j][2], 0x05040100u)};
          *reinterpret_cast<u32x2*>(xw + j * 32 + PHW * 2) = u32x2{__builtin_amdgcn_perm(XS[j][1], XS[j][0], 0x07060302u),
                                                                   __builtin_amdgcn_perm(XS[j][3], XS[j][2], 0x07060302u)};
        } else {
          u32x2* d2 = reinterpret_cast<u32x2*>(xw + j * 64);
          d2[0] = u32x2{XS[j][0], XS[j][1]};
          d2[1] = u32x2{XS[j][2], XS[j][3]};
        }
      }
    };
    auto win_load = [&](int u) { P[u] = *reinterpret_cast<const s16x4*>(xrow + u * 8); };
    auto tap_load = [&](int kk) {
      T[kk] = u32x2{*reinterpret_cast<const unsigned*>(trow + kk * 16), *reinterpret_cast<const unsigned*>(trow + kk * 16 + 8)};
    };
    auto dw_begin = [&]() {
#pragma unroll
      for (int u = 0; u < M - 1 + NKP * WD; ++u) win_load(u);
#pragma unroll
      for (int kk = 0; kk < NKP * WD; ++kk) tap_load(kk);
    };
    auto dw_pass = [&](auto pc) {
      constexpr int p = decltype(pc)::value;
      if constexpr (p + WD < NPASS) {
#if TS_SPLIT_SWITCH_OFF == 2
        if (WM != TS_SPLIT_SWITCH_WM) {
#endif
#pragma unroll
        for (int u = 0; u < NKP; ++u) win_load((p + WD) * NKP + M - 1 + u);
#pragma unroll
        for (int u = 0; u < NKP; ++u) tap_load((p + WD) * NKP + u);
#if TS_SPLIT_SWITCH_OFF == 2
        }
#endif
      }
#pragma unroll
      for (int kk = 0; kk < NKP; ++kk)
#pragma unroll
        for (int m = 0; m < M; ++m) {    // the very first k-step starts from 0 (an inline constant: no register is zeroed)
#if TS_SPLIT_SWITCH_OFF >= 1             // diagnostic builds only (tools/variants.py): the producers' matrix work switched off -- WRONG results, a lower bound
          if (WM == TS_SPLIT_SWITCH_WM && !(p == 0 && kk == 0 && m < M)) continue;       // of what ANY faster depthwise producer could buy those launches
#endif
          d[m] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, T[p * NKP + kk]), P[p * NKP + kk + m],
                                                        (p == 0 && kk == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : d[m], 0, 0, 0);
        }
    };
    auto dw_store = [&](char* dst) {
      // ONE v_cvt_pk_bf16_f32 per pair (the plain cast lowers to two conversions and a v_perm_b32: 36 instead of 12 instructions in
      // this wave's serial chain); hipcc adds no wait states for inline asm, so the last pass's results settle first
      asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const unsigned m0 = pack_bf16_settled(d[m][0], d[m][1]), m1 = pack_bf16_settled(d[m][2], d[m][3]);
        if constexpr (DIL == 2) {
          // lanes q and q ^ 2 hold the even and the odd frames of the same 8-frame group: the even lane stores frames
          // 0..3 (e0 o0 e1 o1), the odd lane frames 4..7 (e2 o2 e3 o3)
          const unsigned t0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m0, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
          const unsigned t1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m1, 0x4E, 0xf, 0xf, true);
          const bool odd = (q >> 1) != 0;
          const unsigned ev = odd ? t1 : m0, od = odd ? m1 : t0;
          *reinterpret_cast<u32x2*>(dst + dw_out[m]) = u32x2{__builtin_amdgcn_perm(od, ev, 0x05040100u), __builtin_amdgcn_perm(od, ev, 0x07060302u)};
        } else {
          *reinterpret_cast<u32x2*>(dst + dw_out[m >> 1] + 8 * (m & 1)) = u32x2{m0, m1};
        }
      }
    };

    unsigned ds = 0;                               // depthwise stages started: selects the tap image
    {
      const int n_main = L.c_in / KC;
      const int n_res = L.c_res / KC;
      const __amdgpu_buffer_rsrc_t rx = rsrc(reinterpret_cast<const char*>(L.x) - TS_GUARD_BYTES);
      const i32x4 rt = raw_rsrc(L.taps_raw, (unsigned)n_main * (4 * TAPB));
      const __amdgpu_buffer_rsrc_t ri = rsrc(n_res ? L.xres : L.x);
      const int lane_i = ((pw * 16 + row) * L.pitch_res + sub * 8) * 2;
      const int chunk_i = KC * L.pitch_res * 2;
      const bool rows2 = ROWS2 && n_main > 0 && !(n_main & 1);

      TilePos dwp;
      dwp.init(tile0, tile_step, a.n_tt, a.n_z);
      int dw_tile = tile0, dw_chunk = 0;
      auto x_origin = [&](const TilePos& p) { return (p.b * L.c_in * a.pitch_in + p.tt * TT - a.padl8) * 2 + TS_GUARD_BYTES; };
      int x_soff = x_origin(dwp);
      auto dw_issue = [&](u32x4 (&XS)[XP]) {
        const bool last_chunk = dw_chunk + 1 == n_main;
        if (last_chunk && dw_tile + tile_step < tile_end) {
          dw_tile += tile_step;
          dwp.advance(a.n_tt, a.n_z);
        }
#pragma unroll
        for (int j = 0; j < XP; ++j) XS[j] = ld16(rx, lane_x + j * 64, x_soff);
        if (last_chunk) {
          dw_chunk = 0;
          x_soff = x_origin(dwp);
        } else {
          ++dw_chunk;
          x_soff += chunk_x;
        }
      };
      int t_next = (n_main > 1 ? 1 : 0) * chunk_t;
      auto tap_dma = [&](int buf, int soff) {        // the whole image of a stage into tapl[buf]
#pragma unroll
        for (int h = 0; h < NTD; ++h) lds_dma16(rt, tapl + buf * TAPB + h * 1024, lane_t, soff + h * 1024);
      };
      auto tap_advance = [&]() { t_next = t_next + chunk_t == n_main * chunk_t ? 0 : t_next + chunk_t; };
      TilePos idp;
      idp.init(tile0, tile_step, a.n_tt, a.n_z);
      int id_tile = tile0, id_s = 0;
      auto i_origin = [&](const TilePos& p) { return (p.b * L.c_res * L.pitch_res + p.tt * TT) * 2; };
      int i_soff = i_origin(idp);
      auto id_issue = [&](u32x4 (&R)[IDP]) {
#pragma unroll
        for (int j = 0; j < IDP; ++j) R[j] = ld16(ri, lane_i + j * 64, i_soff);
        if (++id_s == n_res) {
          id_s = 0;
          if (id_tile + tile_step < tile_end) { id_tile += tile_step; idp.advance(a.n_tt, a.n_z); }
          i_soff = i_origin(idp);
        } else {
          i_soff += chunk_i;
        }
      };

      // prologue: rows and taps of the first stage (a pointwise-only layer has identity stages only: n_main == 0)
      if (rows2) {
        tap_dma(ds & 1, 0);
        dw_issue(X[0]);                              // rows of stages 0 and 1
        dw_issue(X[ROWS2 ? 1 : 0]);
      } else if (n_main) {
        dw_issue(X[0]);
        tap_dma(ds & 1, 0);
      }
      const bool id2 = ID2 && n_res && !(n_res & 1);
      if (n_res) id_issue(I);
      if constexpr (ID2) { if (id2) id_issue(I2); }
      vm_wait<0>();
      auto id_stage = [&](u32x4 (&R)[IDP], bool drain) {
        char* const dst = dwt + (gs & 1) * TILEB;
        if (drain) vm_wait<0>(); else vm_wait<IDP>();
#pragma unroll
        for (int j = 0; j < IDP; ++j) *reinterpret_cast<u32x4*>(dst + (id_out ^ (j << 6))) = R[j];
        id_issue(R);
        stage_barrier();
        ++gs;
      };
      auto stage2 = [&](u32x4 (&XS)[XP]) {
        // rows of this stage: requested two stages ago; its tap image: at the start of the previous stage, BEFORE that stage's row
        // request -- so everything but the XP youngest loads (the next stage's rows) has to be there, and those stay in flight
        // through this stage.  (Identity loads and counter reads issued in between only make the wait stricter; there are never
        // fewer than XP younger operations: dw_issue always issues.)
        char* const dst = dwt + (gs & 1) * TILEB;
        vm_wait<XP>();
        trow = tapl + (ds & 1) * TAPB + tap_off;
        xs_write(XS);
        dw_begin();
        tap_dma((ds + 1) & 1, t_next);                // the next stage's tap image first ...
        dw_issue(XS);                                 // ... then the rows of the stage after next, into the set just consumed
        tap_advance();
        ++ds;
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NPASS>([&](auto pc) { dw_pass(pc); __builtin_amdgcn_sched_barrier(0); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dw_store(dst);
        stage_barrier();
        ++gs;
      };
      for (int tile = tile0; tile < tile_end; tile += tile_step) {
        if (rows2) {
          for (int s = 0; s < n_main; s += 2) {
            stage2(X[0]);
            stage2(X[ROWS2 ? 1 : 0]);
          }
        } else {
          for (int s = 0; s < n_main; ++s, ++gs) {
            char* const dst = dwt + (gs & 1) * TILEB;
            // Everything this wave has in flight -- the rows and the tap image of THIS stage -- was issued at the start of the
            // previous stage: the drain is cheap.
            vm_wait<0>();
            trow = tapl + (ds & 1) * TAPB + tap_off;
            xs_write(X[0]);
            dw_begin();
            dw_issue(X[0]);                               // rows of the next depthwise stage (possibly of the next tile)
            tap_dma((ds + 1) & 1, t_next);                // ... and its tap image, into the buffer the previous stage has finished with
            tap_advance();
            ++ds;
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, NPASS>([&](auto pc) { dw_pass(pc); __builtin_amdgcn_sched_barrier(0); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dw_store(dst);
            stage_barrier();
          }
        }
        if constexpr (ID2) {
          if (id2) {
            // a tile that had depthwise stages drains once (tap DMAs and row loads are in the queue: counted waits cannot separate them);
            // after that only identity loads are in flight and they retire in order
            for (int s = 0; s < n_res; s += 2) {
              id_stage(I, s == 0 && n_main != 0);
              id_stage(I2, false);
            }
          } else {
            for (int s = 0; s < n_res; ++s) id_stage(I, true);
          }
        } else {
          for (int s = 0; s < n_res; ++s) id_stage(I, true);
        }
      }
      vm_wait<0>();                                    // no tap DMA may still be heading for the LDS when the workgroup ends
      stage_barrier();                                 // pairs with the consumers' last stage
    }
    return;
  }

  // ================================= CONSUMER =========================================================
  // v_mfma_f32_16x16x32_bf16: D[t][co] += dw[t][ci] W[co][ci] on 16-frame x 16-channel accumulators, 32 input channels per k-step.  The same
  // matrix-core time as the 32x32x16 form this kernel used through round 3, but the chip holds a visibly higher clock on it (guide: DVFS
  // give-back item 7) and a producer's 8-cycle 4x4x4 products queue behind 16-cycle instead of 32-cycle instructions: the stage loop without
  // global memory takes 1.49 instead of 1.79 us (tools/diag/probe_stage.hip, profiles/round4_probe_stage.txt).
  constexpr int MT16 = 2 * MT, NT16 = 2 * NT;       // 16-frame / 16-channel accumulator tiles per wave
  char* const priv = cons0 + (size_t)wave * ER * EP;
  const int wm = WM == 1 ? 0 : wave / WN, wn = WM == 1 ? wave : wave % WN;
  const int n_c16 = ((a.c_out + 31) >> 5) * 2;      // 16-channel tiles in the packed weights (c_out padded to 32)
  const int kg = lane >> 4;
  const int q4 = (lane >> 2) & 3;
  const int p4 = lane & 3;
  // transposed A-operand reads: lane group kg reads channels 8 kg + q (lo) and + 4 (hi) of the k-step, lane 4q + p supplies frames 4p .. 4p + 3
  int abase[MT16];
#pragma unroll
  for (int mt = 0; mt < MT16; ++mt) abase[mt] = taddr(8 * kg + q4, wm * FW + 16 * mt + 4 * p4);
  constexpr int SEG = FW / 8 <= 16 ? 16 : 32;         // 16-byte segments of an epilogue row (FW frames), padded to a power of two
  constexpr int RPI = 64 / SEG;                       // rows per wave instruction in the epilogue
  const int rsub = lane / SEG, csub = lane % SEG;
  const int lane_w = lane * 16;
  const int lane_y = (rsub * a.pitch_out + csub * 8) * 2;

  // B fragments: TS_SPLIT_RING sets of one k-step each; a slot is refilled right after its last use with the fragment of the k-step RING ahead
  constexpr int RD = TS_SPLIT_RING;
  s16x8 ring[RD][NT16];
  f32x4 acc[MT16][NT16];
  float bnext[NT16];
  s16x8 af[TS_SPLIT_RING == 1 ? MT16 : 3];
  static_assert(TS_SPLIT_RING == 1 || MT16 % 3 == 0, "three rotating A-fragment registers");
  auto read_a1 = [&](const char* src, int ks, int mt) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)src + abase[mt] + ks * 32 * ROWB));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)src + abase[mt] + ks * 32 * ROWB + 4 * ROWB));
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };

  {
    const int n_main = L.c_in / KC;
    const int n_res = L.c_res / KC;
    const int n_stage = n_main + n_res;
    const __amdgpu_buffer_rsrc_t rwm = rsrc(L.pw_w);
    const __amdgpu_buffer_rsrc_t rwr = rsrc(n_res ? L.res_w : L.pw_w);
    const __amdgpu_buffer_rsrc_t ry = rsrc(L.y);
    const __amdgpu_buffer_rsrc_t rse = rsrc(SE ? L.se_y : L.y);
    const unsigned floor2 = (L.relu && !SE) ? 0u : 0x80008000u;        // SE: the ReLU follows the gate + add, in the row-segment stage

    // weight stream: fragments [c_out / 16][c_in / 32][64 lanes][8], walked k-step by k-step over (tile, main stages, residual stages);
    // the pointers below always name the k-step AFTER the one being multiplied
    TilePos wp;
    wp.init(tile0, tile_step, a.n_tt, a.n_z);
    int w_tile = tile0, w_k = 0;
    const int nk_main = 2 * n_main, nk_all = 2 * n_stage;
    __amdgpu_buffer_rsrc_t rwn = rwm;
    int wn_soff[NT16];
    auto w_seek = [&](bool res) {
      rwn = res ? rwr : rwm;
      const int kt = (res ? L.kt_res : L.kt_main) >> 1;
#pragma unroll
      for (int nt = 0; nt < NT16; ++nt) {
        const int c16 = (wp.z * WN + wn) * NT16 + nt;
        wn_soff[nt] = (c16 < n_c16 ? c16 : n_c16 - 1) * kt * 1024;
      }
    };
    auto w_next = [&]() {
#if TS_SPLIT_SWITCH_OFF == 5            // diagnostic build 5: the weight loads are issued but never advance (always the same, cache-hot fragments; WRONG results)
      return;
#endif
      ++w_k;
      if (w_k == nk_all) {
        w_k = 0;
        if (w_tile + tile_step < tile_end) { w_tile += tile_step; wp.advance(a.n_tt, a.n_z); }
        w_seek(nk_main == 0);
      } else if (w_k == nk_main) {
        w_seek(true);
      } else {
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt) wn_soff[nt] += 1024;
      }
    };
    auto load_w = [&](int set, int nt) { ring[set][nt] = __builtin_bit_cast(s16x8, ld16(rwn, lane_w, wn_soff[nt])); };
    auto bias_fetch = [&](const TilePos& p) {
#pragma unroll
      for (int nt = 0; nt < NT16; ++nt) {
        const int col = ((p.z * WN + wn) * NT16 + nt) * 16 + (lane & 15);
        bnext[nt] = L.bias[col < a.c_out ? col : 0];
      }
    };
    // one k-step: MT16 x NT16 products.
#if TS_SPLIT_RING == 1
    // `more_a`: the A fragments of the stage's second k-step replace the first one's as they retire
    auto kstep = [&](const char* src, bool more_a) {
#pragma unroll
      for (int nt = 0; nt < NT16; ++nt) {
#pragma unroll
        for (int mt = 0; mt < MT16; ++mt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], ring[0][nt], acc[mt][nt], 0, 0, 0);
          if (more_a && nt == NT16 - 1) af[mt] = read_a1(src, 1, mt);
        }
#if TS_SPLIT_SWITCH_OFF != 4            // diagnostic build 4: the consumers' weight stream switched off (stale fragments, WRONG results): what the stream costs
        load_w(0, nt);
#endif
      }
      w_next();
    };
#else
    // Frame-tile-major order: the four weight fragments of the k-step stay put while the A fragments stream through THREE registers (the one being
    // multiplied, the next, and the one after that in flight from LDS) instead of six -- the 12 registers that frees, plus slack, hold a SECOND set of
    // weight fragments, so that a fragment is requested two k-steps (a whole stage) before its first use instead of one: the round-6 switch-off
    // runs priced the consumers' weight stream at 16 % of the encoder (10 % even with cache-hot fragments: latency, not bandwidth).
    auto kstep = [&](const char* src, bool more_a) {
      const int set = more_a ? 0 : 1;                      // a stage is two k-steps: the set follows the k-step's parity (static after inlining)
#pragma unroll
      for (int mt = 0; mt < MT16; ++mt) {
#if TS_SPLIT_SWITCH_OFF != 6            // diagnostic build 6: the consumers' A-fragment LDS reads inside the k-step switched off (stale fragments)
        if (mt + 2 < MT16) af[(mt + 2) % 3] = read_a1(src, more_a ? 0 : 1, mt + 2);
        else if (more_a) af[(mt + 2) % 3] = read_a1(src, 1, mt + 2 - MT16);
#endif
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt) {
#if TS_SPLIT_SWITCH_OFF == 7            // diagnostic build 7: the consumers' matrix instructions switched off (operands still fetched and kept alive)
          asm volatile("" :: "v"(af[mt % 3]), "v"(ring[set][nt]));
#else
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt % 3], ring[set][nt], acc[mt][nt], 0, 0, 0);
#endif
#if TS_SPLIT_SWITCH_OFF != 4
          if (mt == MT16 - 1) load_w(set, nt);
#endif
        }
      }
      w_next();
    };
#endif

    TilePos pos;
    pos.init(tile0, tile_step, a.n_tt, a.n_z);
    w_seek(nk_main == 0);
#pragma unroll
    for (int set = 0; set < RD; ++set) {
#pragma unroll
      for (int nt = 0; nt < NT16; ++nt) load_w(set, nt);
      w_next();
    }
    bias_fetch(pos);
    stage_barrier();                                   // stage 0 is in dwt[gs & 1]
    for (int tile = tile0; tile < tile_end; tile += tile_step) {
      const int b = pos.b, t0 = pos.tt * TT;
      const int c16_0 = (pos.z * WN + wn) * NT16;
      const int len_b = a.zero_tail ? a.len[b] : 0;
#pragma unroll
      for (int j = 0; j < NT16; ++j)
#pragma unroll
        for (int i = 0; i < MT16; ++i) acc[i][j] = f32x4{bnext[j], bnext[j], bnext[j], bnext[j]};
      for (int s = 0; s < n_stage; ++s, ++gs) {
        const char* const src = dwt + (gs & 1) * TILEB;
#pragma unroll
        for (int mt = 0; mt < (TS_SPLIT_RING == 1 ? MT16 : 2); ++mt) af[mt] = read_a1(src, 0, mt);
        __builtin_amdgcn_sched_barrier(0);
        kstep(src, true);
        __builtin_amdgcn_sched_barrier(0);
        kstep(src, false);
        stage_barrier();
      }
      // ---- epilogue (the producers are already on the next tile)
      pos.advance_if(tile + tile_step < tile_end, a.n_tt, a.n_z);
      bias_fetch(pos);
      int len_out = 0x7fffffff;
      if (a.zero_tail) len_out = conv_len(len_b, a.kernel, 1, a.padding, a.dilation);
      const int tw = t0 + wm * FW;
      const bool partial = tw + FW > len_out;
      u32x4 keep = u32x4{~0u, ~0u, ~0u, ~0u};
      if (partial) keep = keep_first(keep, len_out - (tw + csub * 8));
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
      const s16x2 f2 = __builtin_bit_cast(s16x2, floor2);
#if TS_SPLIT_DIRECT_STORE
      // experiment: the accumulators leave as they are -- lane (channel n = lane & 15, frame group kg) stores its 4 frames (8 bytes) of every 16-frame
      // tile; a store instruction covers 16 rows x 32 bytes, the six frame tiles of a row follow each other
      if constexpr (!SE) {
        const int lane_d = ((lane & 15) * a.pitch_out + 4 * kg) * 2;
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt) {
          const int ch0 = (c16_0 + nt) * 16;
          const bool chan_ok = ch0 + (lane & 15) < a.c_out;
          const int soff = ((b * a.c_out + ch0) * a.pitch_out + tw) * 2;
#pragma unroll
          for (int mt = 0; mt < MT16; ++mt) {
            unsigned lo = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2,
                pack_bf16_settled(acc[mt][nt][0], acc[mt][nt][1])), f2));
            unsigned hi = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2,
                pack_bf16_settled(acc[mt][nt][2], acc[mt][nt][3])), f2));
            if (partial) {
              const int n = len_out - (tw + 16 * mt + 4 * kg);
              lo &= n >= 2 ? ~0u : (n == 1 ? 0xffffu : 0u);
              hi &= n >= 4 ? ~0u : (n == 3 ? 0xffffu : 0u);
            }
            if (chan_ok) __builtin_amdgcn_raw_buffer_store_b64(u32x2{lo, hi}, ry, lane_d, soff + 32 * mt, 0);
          }
        }
      } else
#endif
      // 32 output channels (two accumulator columns) at a time through the wave-private LDS tile: lane (channel n = lane & 15, frame group
      // kg) writes its 4 frames (8 bytes) of every 16-frame tile into row n (+ 16 for the second column), the rows leave as 16-byte segments
#pragma unroll
      for (int np = 0; np < NT; ++np) {
        const int cob = (c16_0 + 2 * np) * 16;
#pragma unroll
        for (int half = 0; half < 32 / ER; ++half) {
#pragma unroll
          for (int sub = 0; sub < ER / 16; ++sub) {
            const int nt = 2 * np + (ER == 16 ? half : sub);
            char* const prow_w = priv + (size_t)(sub * 16 + (lane & 15)) * EP + 8 * kg;
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt) {
              const unsigned lo = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2,
                  pack_bf16_settled(acc[mt][nt][0], acc[mt][nt][1])), f2));
              const unsigned hi = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2,
                  pack_bf16_settled(acc[mt][nt][2], acc[mt][nt][3])), f2));
              *reinterpret_cast<u32x2*>(prow_w + 32 * mt) = u32x2{lo, hi};
            }
          }
          // the rows leave in batches of four per lane (eight at once are 32 registers on top of the accumulators and the next tile's first weight
          // fragments: the squeeze-excite instantiation spilled 21 registers with them, 12 now)
          constexpr int RB = 4, NB = ER / RPI / RB;
          static_assert(NB * RB * RPI == ER, "row batches");
          if (csub < FW / 8) {
            const int row0 = cob + half * ER;
#if TS_SPLIT_SWITCH_OFF == 10           // diagnostic build 10: every result store of a workgroup lands in the same 16 KB (WRONG results): stores issued and acknowledged, no write traffic
            const int y_soff = ((((b * a.c_out + row0) * a.pitch_out + tw) * 2) & 0x3ff0) + (int)blockIdx.x * 32768;
#else
            const int y_soff = ((b * a.c_out + row0) * a.pitch_out + tw) * 2;
#endif
            const char* const prow = priv + (size_t)rsub * EP + csub * 16;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
              u32x4 v[RB];
#pragma unroll
              for (int i = 0; i < RB; ++i) {
                const u32x2* const pr = reinterpret_cast<const u32x2*>(prow + RPI * (RB * bt + i) * EP);
                v[i] = u32x4{pr[0][0], pr[0][1], pr[1][0], pr[1][1]};
              }
              if constexpr (SE) {
                // four rows at a time, requested here (after the result rows are back in registers): fetching them ahead of the LDS round trip, or
                // a batch ahead, was measured slower -- the epilogue has no registers left for rows in flight (77-147 spilled VGPRs, C3 9.17 ms vs
                // 9.04 with this form and 9.17 without the fusion)
                u32x4 yv[RB];
                float gt[RB];
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                  const int row = row0 + RPI * (RB * bt + i) + rsub;
                  const bool in = row < a.c_out;
                  yv[i] = in ? __builtin_amdgcn_raw_buffer_load_b128(rse, lane_y, y_soff + RPI * (RB * bt + i) * a.pitch_out * 2, 0) : u32x4{0u, 0u, 0u, 0u};
                  gt[i] = in ? L.se_gate[(size_t)b * a.c_out + row] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                  u32x4 o;
#pragma unroll
                  for (int j = 0; j < 4; ++j) {
                    const float lo = fmaf(bf16_lo(yv[i][j]), gt[i], bf16_lo(v[i][j]));
                    const float hi = fmaf(bf16_hi(yv[i][j]), gt[i], bf16_hi(v[i][j]));
                    o[j] = pack_bf16(lo > 0.f ? lo : 0.f, hi > 0.f ? hi : 0.f);
                  }
                  v[i] = o;
                }
              }
#pragma unroll
              for (int i = 0; i < RB; ++i) {
                if (partial) v[i] &= keep;
                if (row0 + RPI * (RB * bt + i) + rsub < a.c_out)
#if TS_SPLIT_SWITCH_OFF == 8            // diagnostic build 8: the result stores switched off
                  asm volatile("" :: "v"(v[i]));
#else
                  if (TS_SPLIT_STORE_AUX != 0 && a.c_out <= 512)      // (wave-uniform) -- see TS_SPLIT_STORE_AUX
                    __builtin_amdgcn_raw_buffer_store_b128(v[i], ry, lane_y, y_soff + RPI * (RB * bt + i) * a.pitch_out * 2, TS_SPLIT_STORE_AUX);
                  else
                    __builtin_amdgcn_raw_buffer_store_b128(v[i], ry, lane_y, y_soff + RPI * (RB * bt + i) * a.pitch_out * 2, 0);
#endif
              }
            }
          }
        }
      }
#if TS_SPLIT_SWITCH_OFF == 9            // diagnostic build 9: the consumers wait for their result stores' acknowledgements before the next tile (results unchanged)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    }
  }
}

template <int NPASS, int XJ, int MT, int WM, int DIL = 1, int NT = 2, bool SE = false>
static int launch_split(SplitArgs& a, hipStream_t stream) {
  constexpr int FW = 32 * MT, TT = FW * WM, CO_WG = 32 * NT * (8 / WM);
  constexpr int ROWB = TT <= 128 ? 256 : 512;
  constexpr int NK_ = NPASS * NKP, CST = (16 * NK_ + 16) % 32 == 16 ? 16 * NK_ + 16 : 16 * NK_ + 32, TAPB = (16 * CST + 1023) / 1024 * 1024;
  a.n_tt = (a.t_out + TT - 1) / TT;
  a.n_z = (round_up(a.c_out, 32) + CO_WG - 1) / CO_WG;
  a.n_tiles = a.batch * a.n_tt * a.n_z;
  const size_t lds = (size_t)2 * KC * ROWB + (size_t)8 * ((WM == 2 || DIL == 2 || MT > 3) ? 16 : 32) * (FW * 2 + 24) +
                     (size_t)4 * (16 * (64 * XJ + 4) * 2 + 2 * TAPB);
  if (lds > 160 * 1024) return TS_EUNSUPPORTED;
  auto kern = tcs_split_kernel<NPASS, XJ, MT, WM, DIL, NT, SE>;
  static bool attr_set[64] = {};                          // per device (one process may drive several GPUs)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TS_EINVAL;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set[dev] = true;
  }
  const int n_cu = cu_count();
  const int grid = a.n_tiles < n_cu ? a.n_tiles : n_cu;
  a.xcd = grid % 8 == 0 ? 1 : 0;
  (void)hipGetLastError();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(768), lds, stream, a);
  return hip_status(hipGetLastError());
}

int launch_split_layer(SplitArgs& a, int npass, int xe, int wm, int dil, hipStream_t stream) {
  if (wm == 4) {
    // wide-frame consumers (round 6): every consumer wave owns 192 frames x 32 channels (workgroup tile 192 x 256) instead of 96 x 64 -- half the
    // weight-fragment loads per matrix instruction (profiles/round6_tcs_256.txt prices that stream at 15 % of the encoder); pointwise-only layers only
    // (with a depthwise stage the 192-frame producer rows and tap images do not fit the LDS beside the 64 KB tile pair)
    if (dil != 1 || npass != 2 || a.layer.c_in != 0) return TS_EUNSUPPORTED;
    if (a.layer.se_y) return a.layer.se_gate ? launch_split<2, 2, 6, 1, 1, 1, true>(a, stream) : TS_EUNSUPPORTED;
    return launch_split<2, 2, 6, 1, 1, 1>(a, stream);
  }
  if (a.layer.se_y) {
    // squeeze-excite tail: the pointwise-only launches of the Citrinet blocks (512 / 256 output channels per workgroup)
    if (!a.layer.se_gate || dil != 1 || npass != 2) return TS_EUNSUPPORTED;
    if (xe == 128 && wm == 1) return launch_split<2, 2, 3, 1, 1, 2, true>(a, stream);
    if (xe == 256 && wm == 2) return launch_split<2, 4, 3, 2, 1, 2, true>(a, stream);
    return TS_EUNSUPPORTED;
  }
  if (wm == 1 && round_up(a.c_out, 32) <= 256) return TS_EUNSUPPORTED;       // narrow layers run on the 192-frame tiles only (split_tile_wm)
#define TS_PIPE(NP_, XJ_, WM_, DIL_) if (npass == NP_ && xe == 64 * XJ_ && wm == WM_ && dil == DIL_) return launch_split<NP_, XJ_, 3, WM_, DIL_>(a, stream);
  TS_PIPE(3, 4, 2, 1) TS_PIPE(4, 4, 2, 1) TS_PIPE(5, 3, 1, 1) TS_PIPE(6, 3, 1, 1) TS_PIPE(7, 3, 1, 1)      /* QuartzNet: K 33..75 */
  TS_PIPE(2, 2, 1, 1) TS_PIPE(3, 3, 1, 1) TS_PIPE(4, 3, 1, 1) TS_PIPE(2, 4, 2, 1)                         /* Citrinet: K 11..41; pointwise only */
  TS_PIPE(8, 5, 1, 2)                                                                                      /* QuartzNet K87, dilation 2 */
#undef TS_PIPE
  return TS_EUNSUPPORTED;
}

// Frames of a time tile for a layer of c_out output channels: 96 (x 512 channels) above 256 channels, else 192 (x 256): the 192-frame tiles have
// the cheaper stage loop.  96 x 256 tiles (two per workgroup, eight consumer waves of 32 channels) were built for the chain launch of round 4 and
// measured again on single launches in round 5: C2 encoder 2.89 against 2.85 ms on the same box (profiles/round5_session2/wm1_ab.log); their instantiations are gone.
int split_tile_wm(int c_out) { return round_up(c_out, 32) > 256 ? 1 : 2; }

}  // namespace ts
