// Split kernel of the fused time-channel-separable sub-block (gfx950): 12 waves = 8 pointwise consumers + 4 depthwise
// producers, 96 / 192-frame tiles, persistent workgroups -- and, since round 4, a whole CHAIN of sub-blocks per launch.
//
//   y_l[b, co, t] = act( sum_ci Wf_l[co, ci] * dw_l[b, ci, t] + bias_l[co] + sum_cr Wr_l[co, cr] * xres_l[b, cr, t] )
//   dw_l[b, ci, t] = sum_u taps_l[ci, u] * x_l[b, ci, t + u - pad],   x_l = y_(l-1) for l >= 1
//
// Replaces the reference's per-sub-block ATen chain (quartznet/blocks.py:166-182 masked_fill + conv1d(groups=C) + masked_fill +
// conv1d(k=1), :222 batch_norm, :332-337 residual add + relu), and with n_layers > 1 the `for layer in self.mconv` loop of a whole
// block (quartznet/blocks.py:317-338) in ONE launch.  DESIGN.md section 3.1 has the measurements behind each step.
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
// Chains (round 4).  Layer l + 1 reads what layer l wrote, and a tile (clip, time tile tt) of layer l + 1 needs the tiles tt - 1,
// tt, tt + 1 of layer l (its own frames + the K - 1 halo).  Every workgroup walks its tiles layer by layer; a consumer wave drains
// its (write-through, sc1) output stores, the eighth one to arrive on an LDS counter adds 1 to the tile's agent-scope counter,
// and the producers of the workgroup that needs the tile read that counter (sc1 load, requested a stage before its first use) in
// front of their first row load -- which is an sc1 load too, so no cache of the reading CU can serve stale bytes
// (cdna_hip_programming.md Guideline 16, R1 with the payload loads as sc1 loads; MI355X_MICROARCH.md "Valid forms", row 1).
// No grid-wide barrier, no assumption about dispatch order or workgroup placement; correctness needs only that every workgroup of
// the launch becomes resident (grid <= compute units, one workgroup per CU) and waits are bounded (status word).
// Deadlock freedom: a workgroup only ever waits for tiles of the PREVIOUS layer, and it enters a wait for layer l's tiles only
// after its producers have handed over every stage of its own layer l-1 tiles (the barrier that ends a layer comes first), so
// its own layer l-1 epilogues complete without it; by induction over the layers every wait ends.
#include "tcs_shared.hpp"

namespace ts {

namespace {

constexpr unsigned SPIN_LIMIT = 1u << 21;

typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ unsigned flag_load(const unsigned* p) {
  return __hip_atomic_load((const gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the tiles (b, tt - 1 .. tt + 1) of the previous layer: lanes 0..2 hold one counter each (index < 0: lane takes no part)
struct TileWait {
  int idx;
  __device__ __forceinline__ void aim(int b, int tt, int n_tt, int lane) {
    const int t = tt - 1 + lane;
    idx = (lane < 3 && t >= 0 && t < n_tt) ? b * n_tt + t : -1;
  }
  __device__ __forceinline__ unsigned peek(const unsigned* layer_flags, unsigned need) const {
    return idx >= 0 ? flag_load(layer_flags + idx) : need;
  }
  // blocking form: poll (relaxed, one wave-instruction per round) until all three counters have reached `need`
  __device__ __forceinline__ void block(const unsigned* layer_flags, unsigned need, unsigned* status, int lane) const {
    for (unsigned spins = 0;; ++spins) {
      const unsigned v = peek(layer_flags, need);
      if (__all(v >= need)) break;
      if (spins > SPIN_LIMIT) {                       // a workgroup of the launch never became resident: give up loudly
        if (lane == 0) __hip_atomic_store((gu32*)status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     // compiler-only: no load of the tile may move above the poll
  }
};

}  // namespace

// DIL == 2 (dilation-2 layers, K87 of QuartzNet): the even and the odd frames of a row are two independent dilation-1
// sequences (y[2s+p] = sum_u w[u] x[2(s+u)+p - pad], pad even).  The producers stage each row as [even | odd] halves,
// lane runs 0,1 filter the even half and 2,3 the odd half with dilation-1 tap fragments (no zero-stuffed Toeplitz rows:
// 24 k-steps instead of 45), and lane pairs re-interleave their results on the way into the dw tile.
// NT: 32-channel output tiles per consumer wave -- 2: a workgroup covers 512 (WM = 1) or 256 (WM = 2) output channels; 1 (WM = 1 only): 96 frames x
// 256 channels, for layers of at most 256 output channels whose 192-frame tiling would leave a compute unit a single tile per layer (nothing
// to overlap its prologue and epilogue with).
// CHAIN: false -- ONE layer, known at compile time (no counters, plain loads and stores: the instantiation every single-layer launch takes);
// true -- a.n_layers layers, sc1 activations, published tiles.  Two instantiations because the chain's bookkeeping costs registers in a kernel
// that has none to spare: with the layer count a runtime value the single-layer launches of QuartzNet15x5 ran 6 % slower (11 spilled VGPRs in
// the producers' stage loop), measured on one box.
// SE: the epilogue closes a CitrinetBlock -- y = relu(gate[b][co] * se_y[b][co][t] + result) with the main branch's output se_y read as 16-byte row
// segments right where the result rows leave (single-layer launches only; what ts_se_apply_fwd did in a separate pass over three tensors).
template <int NPASS, int XJ, int MT, int WM, int DIL = 1, int NT = 2, bool CHAIN = false, bool SE = false>
__global__ __launch_bounds__(768) void tcs_split_kernel(const ChainArgs a) {
  static_assert(!(SE && CHAIN), "the squeeze-excite tail exists for single-layer launches");
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
  constexpr int ER = (WM == 2 || DIL == 2) ? 16 : 32;   // output-channel rows of a consumer's epilogue tile
  constexpr int PHW = 32 * XJ;                    // DIL == 2: frames of a staged half row
  constexpr int AUX_SC1 = CHAIN ? 16 : 0;         //                     // cache-policy bit sc1 of the buffer instructions: write-through stores, L1-bypassing loads

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const dwt = smem;                                             // [2][KC][ROWB]
  char* const cons0 = smem + 2 * TILEB;                               // [8][ER][EP] epilogue tiles
  char* const prod0 = cons0 + 8 * ER * EP;                            // [4][XSB + 2 * TAPB]
  unsigned* const arrive = reinterpret_cast<unsigned*>(prod0 + 4 * (XSB + 2 * TAPB));   // consumer waves done with their stores
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
  auto taddr = [](int c, int t) { return c * ROWB + ((((t >> 3) ^ ((c & 3) * 5))) << 4) + ((t & 7) << 1); };
  constexpr int RSRC_FLAGS = 0x00020000;
  auto rsrc = [](const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, RSRC_FLAGS); };
  auto ld16 = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); };
  // activations another workgroup of this launch may have written: every load of them bypasses this CU's L1
  auto ld16_sc1 = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX_SC1)); };
  const int nct = a.batch * a.n_tt;               // counters per layer
  unsigned* const status = a.flags;               // word 0; the counters start 16 bytes in
  unsigned* const counters = a.flags + 4;
  const int N_LAYERS = CHAIN ? a.n_layers : 1;
  unsigned gs = 0;
  if (tid == 0) *arrive = 0;                      // ordered before its first use by the barrier that opens layer 0

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
    int dw_out[M];
#pragma unroll
    for (int m = 0; m < M; ++m)
      dw_out[m] = DIL == 2 ? taddr(pw * 16 + row, (q & 1) * 2 * RUN + 8 * m + 4 * (q >> 1)) : taddr(pw * 16 + row, q * RUN + 4 * m);
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
    // identity rows in flight: one stage ahead; TWO (a second register set, alternating statically) in the single-layer instantiation when the
    // residual has an even number of stages -- an identity stage is as short as the consumers' k-loop, shorter than a loaded HBM round trip.
    // The chain instantiation has no registers for the second set (it cost 20 spilled VGPRs in the stage loop there).
    constexpr bool ID2 = !CHAIN && WM == 1;
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
#pragma unroll
        for (int u = 0; u < NKP; ++u) win_load((p + WD) * NKP + M - 1 + u);
#pragma unroll
        for (int u = 0; u < NKP; ++u) tap_load((p + WD) * NKP + u);
      }
#pragma unroll
      for (int kk = 0; kk < NKP; ++kk)
#pragma unroll
        for (int m = 0; m < M; ++m)      // the very first k-step starts from 0 (an inline constant: no register is zeroed)
          d[m] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, T[p * NKP + kk]), P[p * NKP + kk + m],
                                                        (p == 0 && kk == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : d[m], 0, 0, 0);
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
          *reinterpret_cast<u32x2*>(dst + dw_out[m]) = u32x2{m0, m1};
        }
      }
    };

    unsigned ds = 0;                               // depthwise stages started: selects the tap image
    for (int l = 0; l < N_LAYERS; ++l) {
      const ChainLayer& L = a.layer[l];
      const int n_main = L.c_in / KC;
      const int n_res = L.c_res / KC;
      const __amdgpu_buffer_rsrc_t rx = rsrc(reinterpret_cast<const char*>(L.x) - TS_GUARD_BYTES);
      const i32x4 rt = raw_rsrc(L.taps_raw, (unsigned)n_main * (4 * TAPB));
      const __amdgpu_buffer_rsrc_t ri = rsrc(n_res ? L.xres : L.x);
      const int lane_i = ((pw * 16 + row) * L.pitch_res + sub * 8) * 2;
      const int chunk_i = KC * L.pitch_res * 2;
      const bool rows2 = ROWS2 && n_main > 0 && !(n_main & 1);
      // tiles of the previous layer this layer's row loads wait for
      const unsigned* const wflag = (CHAIN && L.wait_in) ? counters + (size_t)(l - 1) * nct : nullptr;
      const unsigned need = (unsigned)a.n_z;
      TileWait tw;
      unsigned tw_seen = need;
      bool tw_pending = false;

      TilePos dwp;
      dwp.init(tile0, tile_step, a.n_tt, a.n_z);
      int dw_tile = tile0, dw_chunk = 0;
      auto x_origin = [&](const TilePos& p) { return (p.b * L.c_in * a.pitch_in + p.tt * TT - a.padl8) * 2 + TS_GUARD_BYTES; };
      int x_soff = x_origin(dwp);
      auto dw_issue = [&](u32x4 (&XS)[XP]) {
        if (tw_pending) {                            // first rows of a new tile: its counters were requested a stage ago
          if (!__all(tw_seen >= need)) tw.block(wflag, need, status, lane);
          tw_pending = false;
        }
        const bool last_chunk = dw_chunk + 1 == n_main;
        if (last_chunk && dw_tile + tile_step < tile_end) {
          dw_tile += tile_step;
          dwp.advance(a.n_tt, a.n_z);
          if (wflag) {                               // the counters of the NEXT tile, in front of this stage's row loads
            tw.aim(dwp.b, dwp.tt, a.n_tt, lane);
            tw_seen = tw.peek(wflag, need);
            tw_pending = true;
          }
        }
#pragma unroll
        for (int j = 0; j < XP; ++j) XS[j] = ld16_sc1(rx, lane_x + j * 64, x_soff);
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

      // prologue of the layer: rows and taps of its first stage (a pointwise-only layer has identity stages only: n_main == 0)
      if (wflag && n_main) {
        tw.aim(dwp.b, dwp.tt, a.n_tt, lane);
        tw.block(wflag, need, status, lane);
      }
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
      vm_wait<0>();                                    // no tap DMA may still be heading for the LDS when the layer (the workgroup) ends
      stage_barrier();                                 // pairs with the consumers' last stage of the layer
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
  const int rsub = lane >> 4, csub = lane & 15;
  const int lane_w = lane * 16;
  const int lane_y = (rsub * a.pitch_out + csub * 8) * 2;

  s16x8 ring[NT16];                                  // B fragments of ONE k-step; a slot is refilled for the next k-step right after its last use
  f32x4 acc[MT16][NT16];
  float bnext[NT16];
  s16x8 af[MT16];
  auto read_a1 = [&](const char* src, int ks, int mt) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)src + abase[mt] + ks * 32 * ROWB));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)src + abase[mt] + ks * 32 * ROWB + 4 * ROWB));
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };

  for (int l = 0; l < N_LAYERS; ++l) {
    const ChainLayer& L = a.layer[l];
    const int n_main = L.c_in / KC;
    const int n_res = L.c_res / KC;
    const int n_stage = n_main + n_res;
    const __amdgpu_buffer_rsrc_t rwm = rsrc(L.pw_w);
    const __amdgpu_buffer_rsrc_t rwr = rsrc(n_res ? L.res_w : L.pw_w);
    const __amdgpu_buffer_rsrc_t ry = rsrc(L.y);
    const __amdgpu_buffer_rsrc_t rse = rsrc(SE ? L.se_y : L.y);
    const unsigned floor2 = (L.relu && !SE) ? 0u : 0x80008000u;        // SE: the ReLU follows the gate + add, in the row-segment stage
    const bool publish = CHAIN && l + 1 < N_LAYERS;           // a later layer of this launch reads y
    unsigned* const pflag = counters + (size_t)l * nct;

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
    auto load_w = [&](int nt) { ring[nt] = __builtin_bit_cast(s16x8, ld16(rwn, lane_w, wn_soff[nt])); };
    auto bias_fetch = [&](const TilePos& p) {
#pragma unroll
      for (int nt = 0; nt < NT16; ++nt) {
        const int col = ((p.z * WN + wn) * NT16 + nt) * 16 + (lane & 15);
        bnext[nt] = L.bias[col < a.c_out ? col : 0];
      }
    };
    // one k-step: NT16 x MT16 products; `more_a`: the A fragments of the stage's second k-step replace the first one's as they retire
    auto kstep = [&](const char* src, bool more_a) {
#pragma unroll
      for (int nt = 0; nt < NT16; ++nt) {
#pragma unroll
        for (int mt = 0; mt < MT16; ++mt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], ring[nt], acc[mt][nt], 0, 0, 0);
          if (more_a && nt == NT16 - 1) af[mt] = read_a1(src, 1, mt);
        }
        load_w(nt);
      }
      w_next();
    };

    int pub_idx = -1;                                  // tile whose publication is pending (counter index), or -1
    auto publish_tile = [&](unsigned* flag) {
      unsigned old = 0;
      if (lane == 0) old = __hip_atomic_fetch_add((TS_LDS unsigned*)arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      old = __builtin_amdgcn_readfirstlane(old);
      if ((old & 7u) == 7u && lane == 0) __hip_atomic_fetch_add((gu32*)flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    TilePos pos;
    pos.init(tile0, tile_step, a.n_tt, a.n_z);
    w_seek(nk_main == 0);
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt) load_w(nt);
    w_next();
    bias_fetch(pos);
    stage_barrier();                                   // stage 0 of the layer is in dwt[gs & 1]
    for (int tile = tile0; tile < tile_end; tile += tile_step) {
      const int b = pos.b, t0 = pos.tt * TT, tt = pos.tt;
      const int c16_0 = (pos.z * WN + wn) * NT16;
      const int len_b = a.zero_tail ? a.len[b] : 0;
#pragma unroll
      for (int j = 0; j < NT16; ++j)
#pragma unroll
        for (int i = 0; i < MT16; ++i) acc[i][j] = f32x4{bnext[j], bnext[j], bnext[j], bnext[j]};
      for (int s = 0; s < n_stage; ++s, ++gs) {
        const char* const src = dwt + (gs & 1) * TILEB;
#pragma unroll
        for (int mt = 0; mt < MT16; ++mt) af[mt] = read_a1(src, 0, mt);
        __builtin_amdgcn_sched_barrier(0);
        kstep(src, true);
        __builtin_amdgcn_sched_barrier(0);
        kstep(src, false);
        if (pub_idx >= 0) {
          // the previous tile's stores are older than the 2 NT16 weight loads of this stage: a counted wait covers exactly them
          vm_wait<2 * NT16>();
          publish_tile(pflag + pub_idx);
          pub_idx = -1;
        }
        stage_barrier();
      }
      // ---- epilogue (the producers are already on the next tile, or on the next layer's first stage)
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
          if (csub < FW / 8) {
            const int row0 = cob + half * ER;
            const int y_soff = ((b * a.c_out + row0) * a.pitch_out + tw) * 2;
            const char* const prow = priv + (size_t)rsub * EP + csub * 16;
            u32x4 v[ER / 4];
#pragma unroll
            for (int i = 0; i < ER / 4; ++i) {
              const u32x2* const pr = reinterpret_cast<const u32x2*>(prow + 4 * i * EP);
              v[i] = u32x4{pr[0][0], pr[0][1], pr[1][0], pr[1][1]};
            }
            if constexpr (SE) {
              // four rows at a time, requested here (after the result rows are back in registers): fetching them ahead of the LDS round trip, or
              // a batch ahead, was measured slower -- the epilogue has no registers left for rows in flight (77-147 spilled VGPRs, C3 9.17 ms vs
              // 9.04 with this form and 9.17 without the fusion)
#pragma unroll
              for (int i0 = 0; i0 < ER / 4; i0 += 4) {
                u32x4 yv[4];
                float gt[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const int row = row0 + 4 * (i0 + i) + rsub;
                  const bool in = row < a.c_out;
                  yv[i] = in ? __builtin_amdgcn_raw_buffer_load_b128(rse, lane_y, y_soff + 4 * (i0 + i) * a.pitch_out * 2, 0) : u32x4{0u, 0u, 0u, 0u};
                  gt[i] = in ? L.se_gate[(size_t)b * a.c_out + row] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  u32x4 o;
#pragma unroll
                  for (int j = 0; j < 4; ++j) {
                    const float lo = fmaf(bf16_lo(yv[i][j]), gt[i], bf16_lo(v[i0 + i][j]));
                    const float hi = fmaf(bf16_hi(yv[i][j]), gt[i], bf16_hi(v[i0 + i][j]));
                    o[j] = pack_bf16(lo > 0.f ? lo : 0.f, hi > 0.f ? hi : 0.f);
                  }
                  v[i0 + i] = o;
                }
              }
            }
#pragma unroll
            for (int i = 0; i < ER / 4; ++i) {
              if (partial) v[i] &= keep;
              // write-through (sc1): what a later layer of this launch reads must not sit dirty in this XCD's L2
              if (row0 + 4 * i + rsub < a.c_out)
                __builtin_amdgcn_raw_buffer_store_b128(v[i], ry, lane_y, y_soff + 4 * i * a.pitch_out * 2, AUX_SC1);
            }
          }
        }
      }
      if (publish) {
        // Publication (R1): every storing wave drains its stores, then checks in on an LDS counter -- not the stage barrier: the producers
        // may already be waiting for this very tile, and the barrier needs them; the last of the eight to arrive signals for the tile.
        // A tile that is followed by another one of this layer publishes one stage LATE (behind the first stage of the next tile, below):
        // its write-through stores then complete under that stage's products instead of stalling the wave here, and with two tiles per
        // workgroup nobody needs the tile that early.  The last tile of a layer publishes at once.
        if (tile + tile_step < tile_end) {
          pub_idx = b * a.n_tt + tt;
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          publish_tile(pflag + b * a.n_tt + tt);
        }
      }
    }
  }
}

template <int NPASS, int XJ, int MT, int WM, int DIL = 1, int NT = 2, bool SE = false>
static int launch_split(ChainArgs& a, hipStream_t stream) {
  const bool chain = a.n_layers > 1;
  if (SE && chain) return TS_EUNSUPPORTED;
  constexpr int FW = 32 * MT, TT = FW * WM, CO_WG = 32 * NT * (8 / WM);
  constexpr int ROWB = TT <= 128 ? 256 : 512;
  constexpr int NK_ = NPASS * NKP, CST = (16 * NK_ + 16) % 32 == 16 ? 16 * NK_ + 16 : 16 * NK_ + 32, TAPB = (16 * CST + 1023) / 1024 * 1024;
  a.n_tt = (a.t_out + TT - 1) / TT;
  a.n_z = (round_up(a.c_out, 32) + CO_WG - 1) / CO_WG;
  a.n_tiles = a.batch * a.n_tt * a.n_z;
  const size_t lds = (size_t)2 * KC * ROWB + (size_t)8 * ((WM == 2 || DIL == 2) ? 16 : 32) * (FW * 2 + 24) +
                     (size_t)4 * (16 * (64 * XJ + 4) * 2 + 2 * TAPB) + 16;
  if (lds > 160 * 1024) return TS_EUNSUPPORTED;
  auto kern = SE ? tcs_split_kernel<NPASS, XJ, MT, WM, DIL, NT, false, SE>
                 : (chain ? tcs_split_kernel<NPASS, XJ, MT, WM, DIL, NT, true> : tcs_split_kernel<NPASS, XJ, MT, WM, DIL, NT, false>);
  static bool attr_set[2][64] = {};                       // per device (one process may drive several GPUs)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TS_EINVAL;
  if (!attr_set[chain][dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set[chain][dev] = true;
  }
  const int n_cu = cu_count();
  const int grid = a.n_tiles < n_cu ? a.n_tiles : n_cu;
  a.xcd = grid % 8 == 0 ? 1 : 0;
  if (chain) {
    // counters + status word of this launch: a memset node of its own, replayed with the launch
    const size_t bytes = (size_t)round_up((4 + a.n_layers * a.batch * a.n_tt) * 4, 16);
    hipError_t e = hipMemsetAsync(a.flags, 0, bytes, stream);
    if (e != hipSuccess) return (int)e;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(768), lds, stream, a);
  return hip_status(hipGetLastError());
}

int launch_split_chain(ChainArgs& a, int npass, int xe, int wm, int dil, hipStream_t stream) {
  if (a.layer[0].se_y) {
    // squeeze-excite tail: the pointwise-only launches of the Citrinet blocks (512 / 256 output channels per workgroup)
    if (a.n_layers != 1 || !a.layer[0].se_gate || dil != 1 || npass != 2) return TS_EUNSUPPORTED;
    if (xe == 128 && wm == 1) return launch_split<2, 2, 3, 1, 1, 2, true>(a, stream);
    if (xe == 256 && wm == 2) return launch_split<2, 4, 3, 2, 1, 2, true>(a, stream);
    return TS_EUNSUPPORTED;
  }
  if (wm == 1 && dil == 1 && round_up(a.c_out, 32) <= 256) {
    // narrow layers on 96-frame tiles (split_tile_rows() chose them): 8 consumer waves x 32 output channels
#define TS_PIPE1(NP_, XJ_) if (npass == NP_ && xe == 64 * XJ_) return launch_split<NP_, XJ_, 3, 1, 1, 1>(a, stream);
    TS_PIPE1(3, 3) TS_PIPE1(4, 3) TS_PIPE1(2, 2) TS_PIPE1(2, 3)
#undef TS_PIPE1
    return TS_EUNSUPPORTED;
  }
#define TS_PIPE(NP_, XJ_, WM_, DIL_) if (npass == NP_ && xe == 64 * XJ_ && wm == WM_ && dil == DIL_) return launch_split<NP_, XJ_, 3, WM_, DIL_>(a, stream);
  TS_PIPE(3, 4, 2, 1) TS_PIPE(4, 4, 2, 1) TS_PIPE(5, 3, 1, 1) TS_PIPE(6, 3, 1, 1) TS_PIPE(7, 3, 1, 1)      /* QuartzNet: K 33..75 */
  TS_PIPE(2, 2, 1, 1) TS_PIPE(3, 3, 1, 1) TS_PIPE(4, 3, 1, 1) TS_PIPE(2, 4, 2, 1)                         /* Citrinet: K 11..41; pointwise only */
  TS_PIPE(8, 5, 1, 2)                                                                                      /* QuartzNet K87, dilation 2 */
#undef TS_PIPE
  return TS_EUNSUPPORTED;
}

// Frames of a time tile for a layer of c_out output channels: 96 (x 512 channels, or x 256 with one 32-channel tile per consumer wave), or 192
// (x 256).  The 192-frame tiles have the cheaper stage loop and are what every single-layer launch takes.  A CHAIN whose 192-frame grid gives a
// workgroup a single tile per layer waits, at every layer boundary, for its own epilogue and its neighbours' (QuartzNet's 256-channel blocks at
// 64 x 751 frames: 256 tiles on 256 compute units, 110 -> 123 us per block); on 96-frame tiles it has two, and the hand-over is free.
int split_tile_wm(int c_out, int batch, int t_out, bool chain) {
  if (round_up(c_out, 32) > 256) return 1;
  if (!chain) return 2;
  const int n192 = batch * ((t_out + 191) / 192), n96 = batch * ((t_out + 95) / 96);
  const int n_cu = cu_count();
  return (n192 < 2 * n_cu && n96 >= 2 * n_cu) ? 1 : 2;
}

}  // namespace ts

extern "C" int64_t ts_tcs_chain_workspace_bytes(int32_t batch, int32_t t_out, int32_t n_layers) {
  if (batch <= 0 || t_out <= 0 || n_layers <= 0) return 0;
  const int64_t n_tt = (t_out + 95) / 96;              // the finer of the two tile grids
  return ((4 + (int64_t)n_layers * batch * n_tt) * 4 + 15) / 16 * 16;
}

extern "C" int ts_tcs_chain_fwd(const ts_tcs_desc* descs, int32_t n_layers, const void* const* x, const void* const* x_res,
                                void* const* y, const int32_t* len, void* workspace, int64_t workspace_bytes, void* stream_) {
  using namespace ts;
  if (!descs || !x || !y || !len || n_layers < 1) return TS_EINVAL;
  if (n_layers > TS_TCS_CHAIN_MAX) return TS_EUNSUPPORTED;
  const ts_tcs_desc& d0 = descs[0];
  if (d0.batch <= 0 || d0.c_out <= 0 || d0.t_out <= 0) return TS_EINVAL;
  if (n_layers > 1 && (!workspace || workspace_bytes < ts_tcs_chain_workspace_bytes(d0.batch, d0.t_out, n_layers))) return TS_EINVAL;
  if (reinterpret_cast<uintptr_t>(workspace) % 16) return TS_EINVAL;
  const int both = TS_TCS_IN_TAILZERO | TS_TCS_OUT_ZERO_TAIL;
  ChainArgs a{};
  for (int l = 0; l < n_layers; ++l) {
    const ts_tcs_desc& d = descs[l];
    if (!x[l] || !y[l] || !d.pw_w || !d.bias) return TS_EINVAL;
    if (d.c_res > 0 && (!x_res || !x_res[l] || !d.res_w)) return TS_EINVAL;
    if (!d.dw_taps_raw || !d.pw_w16 || (d.c_res > 0 && !d.res_w16)) return TS_EUNSUPPORTED;
    const bool same = d.batch == d0.batch && d.c_out == d0.c_out && d.t_in == d0.t_out && d.t_out == d0.t_out && d.pitch_in == d0.pitch_in &&
                      d.pitch_out == d0.pitch_in && d.kernel == d0.kernel && d.padding == d0.padding && d.dw_ksteps == d0.dw_ksteps;
    const bool shape = d.depthwise && d.stride == 1 && d.dilation == 1 && !d.out_fp32 && (d.flags & both) == both && !(d.flags & TS_TCS_TAPS_PHASE) &&
                       d.kernel == 2 * d.padding + 1 && d.c_in > 0 && d.c_in % KC == 0 && d.c_res % KC == 0 && d.dw_ksteps % NKP == 0 &&
                       (d.c_res == 0 || d.res_stride <= 1);
    if (!same || !shape) return TS_EUNSUPPORTED;
    if (l > 0 && (x[l] != y[l - 1] || d.c_in != d0.c_out)) return TS_EUNSUPPORTED;
    for (int j = 0; j <= l; ++j)                       // a residual input written inside the launch would need its own wait
      if (d.c_res > 0 && x_res[l] == y[j]) return TS_EUNSUPPORTED;
    // 32-bit byte offsets inside the buffer descriptors
    const int64_t cmax = d.c_in > d.c_out ? d.c_in : d.c_out;
    if ((int64_t)d.batch * cmax * d.pitch_in * 2 + TS_GUARD_BYTES >= (1ll << 31)) return TS_EUNSUPPORTED;
    if (d.c_res > 0 && (int64_t)d.batch * d.c_res * d.pitch_res * 2 >= (1ll << 31)) return TS_EUNSUPPORTED;
    ChainLayer& L = a.layer[l];
    L.x = static_cast<const unsigned short*>(x[l]);
    L.xres = d.c_res > 0 ? static_cast<const unsigned short*>(x_res[l]) : nullptr;
    L.y = static_cast<unsigned short*>(y[l]);
    L.taps_raw = static_cast<const unsigned short*>(d.dw_taps_raw);
    L.pw_w = static_cast<const unsigned short*>(d.pw_w16);
    L.res_w = static_cast<const unsigned short*>(d.res_w16);
    L.bias = d.bias;
    L.c_in = d.c_in; L.c_res = d.c_res; L.pitch_res = d.c_res > 0 ? d.pitch_res : d.pitch_in; L.relu = d.relu;
    L.kt_main = round_up(d.c_in, KC) / 16;
    L.kt_res = round_up(d.c_res > 0 ? d.c_res : 1, KC) / 16;
    L.wait_in = l > 0 ? 1 : 0;
  }
  a.len = len;
  a.flags = static_cast<unsigned*>(workspace);
  a.n_layers = n_layers;
  a.batch = d0.batch; a.c_out = d0.c_out; a.pitch_in = d0.pitch_in; a.pitch_out = d0.pitch_out; a.t_out = d0.t_out;
  a.kernel = d0.kernel; a.padding = d0.padding; a.dilation = 1;
  a.zero_tail = 1;
  const int npass = d0.dw_ksteps / NKP;
  if (npass > 7) return TS_EUNSUPPORTED;
  const int padl4 = round_up(d0.padding, 4);
  a.padl8 = round_up(padl4, 8);
  a.woff = a.padl8 - padl4;
  const int WM = split_tile_wm(d0.c_out, d0.batch, d0.t_out, n_layers > 1);
  const int TTp = 96 * WM;
  const int n_ttp = (d0.t_out + TTp - 1) / TTp;
  const int xe = round_up(a.woff + TTp + 4 * d0.dw_ksteps, 64);
  bool fits = (n_ttp - 1) * TTp - a.padl8 + xe <= d0.pitch_in && d0.pitch_in - d0.t_in >= a.padl8 && d0.pitch_out >= n_ttp * TTp;
  for (int l = 0; l < n_layers; ++l)
    if (descs[l].c_res > 0 && descs[l].pitch_res < (n_ttp - 1) * TTp + round_up(TTp, 64)) fits = false;
  if (!fits) return TS_EUNSUPPORTED;
  return launch_split_chain(a, npass, xe, WM, 1, reinterpret_cast<hipStream_t>(stream_));
}
