// Token-major bf16 GEMM with fused epilogue for gfx950 (MI355X): the linears and convolutions of the wav2vec2 encoder
// (transformers.Wav2Vec2Model reached from huggingface/compatibility.py:31-42: feature projection, q / k / v, out_proj, the two
// feed-forward linears, conv layers 1-6 as overlapping-row products), replacing the hipBLASLt / rocBLAS calls of rounds 1-2.
//
//   y[m][n] = act( sum_k x[m][k] w[n][k] + bias[n] ) + res[m][n]          x: bf16 rows of pitch lda, w: bf16 [n][k] (nn.Linear layout)
//
// Both operands are K-contiguous rows ("NT" product), so A and B tiles are staged the same way and fragments of
// v_mfma_f32_16x16x32_bf16 are plain 16-byte LDS reads (that shape: it sustains a higher clock than 32x32x16 under load, MI355X_MICROARCH.md).
//  * 256 x 256 output tile per workgroup, 8 waves as 2 (rows) x 4 (columns): a wave owns 128 x 64 = 8 x 4 accumulator blocks
//    of 16 x 16 (128 VGPRs); per 32-deep half-stage 12 fragment reads feed 32 MFMAs.  The big tile is what keeps the operand stream under the
//    L2 -> CU bandwidth: 32 KiB per 32-deep half-stage for 1 024 MFMA cycles per SIMD = 32 B / clk / CU.
//  * operands travel global -> LDS by DMA (buffer_load ... lds, 1 KiB per wave-instruction) into a ring of FOUR half-stages
//    (32 deep: A 16 KiB + B 16 KiB each, 128 KiB in all); up to three are in flight while one is multiplied, counted vmcnt waits.
//    Rows are 64 bytes; the 16-byte chunk index is XOR-ed with {0, 3, 2, 1}[(row >> 2) & 3] -- on the SOURCE address, the DMA's LDS image is
//    lane-linear -- so that any 16 consecutive rows of a fragment read cover all 64 banks.  (Register staging -- 16-byte loads two
//    half-stages ahead + ds_write_b128 -- was built as well and measured 15-25 % slower on every shape.)
//  * ping-pong: a wave alternates a LOAD phase (fragment reads, DMA issue) with an MFMA phase of 16 back-to-back MFMAs, one barrier
//    per phase, and the second wave row runs one phase behind the first: on every SIMD one wave multiplies while the other loads.
//  * epilogue: bias and erf-GELU in registers, then staged through the idle ring for 16-byte row stores: residual add (f32), f32 and /
//    or bf16 result; rows beyond M are clamped on the way in and
//    masked on the way out; N % 32 == 0 and K % 32 == 0 (every linear of the supported checkpoints).
//  * workgroup -> tile map keeps the column tiles of one row panel on one XCD (they share the A rows in that XCD's L2).
#include "ts_common.hpp"

#include <cstdlib>

namespace ts {

namespace {

constexpr int GM = 256, GN = 256, GKH = 32;      // tile, half-stage depth
constexpr int GRING = 4;
constexpr int GROWB = GKH * 2;                   // 64-byte LDS rows
constexpr int GHALFB = GM * GROWB;               // 16 KiB: one operand tile of a half-stage
constexpr int GSTAGEB = 2 * GHALFB;              // 32 KiB
constexpr int GEMM_LDS = 8 * 128 * 144;          // the ring (4 x 32 KiB) or the bf16 epilogue's 8 tiles of 128 rows x 144 bytes

struct GemmArgs {
  const unsigned short* x;
  const unsigned short* w;
  const unsigned short* wf;                      // PB kernels: w as MFMA B fragments [N / 16][K / 32][64 lanes][8] (gemm_nt_pack_w)
  const float* bias;
  const float* res;
  float* y;
  unsigned short* y16;
  long long lda, ldw, ld_res, ldc, ld16;
  long long sx, sy;                              // batch strides (elements) of x and of y / y16 / res
  long long sw;                                  // batch stride of w (0: shared by the batch; split-K: both operands step along the contraction)
  int M, N, K, act;                              // act bit 0: GELU
  int n_mt, n_nt;
  int blk_c;                                      // column tiles per block of the tile walk (divides n_nt)
};

template <int N>
__device__ __forceinline__ void vmw() {
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, below f32 GELU noise; a third of the instructions of ocml's erff) -- the same
// form as the other GELU sites of the encoder (csrc/w2v_enc.hip)
__device__ __forceinline__ float erf_as_g(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float r = 1.f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_g(float x) { return 0.5f * x * (1.f + erf_as_g(x * 0.70710678118654752f)); }

}  // namespace

// PB: the B operand does not pass through LDS.  The weights are static, so they are packed once (gemm_nt_pack_w) in the fragment order
// of v_mfma_f32_16x16x32_bf16 -- 1 KiB per (16 columns, 32-deep step), lane-linear -- and every wave loads the four fragments of its 64
// columns straight into a register ring three steps deep (perfectly coalesced 1-KiB loads, L2-resident: all row tiles share them).
// That takes a quarter of the bytes off the LDS pipe (96 + 32 KiB per step were exactly the 128 B / clk it can move in a step's 1 024 MFMA
// cycles), halves the LDS-DMA instructions -- whose issue cost is what the ring showed (tools/diag/gemm_exp.py: 21 % of the 8192^3
// product with L2-hot sources) -- and leaves the MFMA phase free of memory instructions.
// WMR: wave rows.  2: the 256 x 256 tile above, 8 waves.  1 (packed weights only): a 128 x 256 tile on 4 waves and 72 KiB of LDS, so that TWO
// workgroups share a CU: they fall out of step by themselves, one's epilogue and cold prologue run under the other's k-loop, and the matrix
// core is handed back and forth between the two waves of a SIMD without the stagger barriers.
template <bool PB, int WMR = 2>
__global__ __launch_bounds__(256 * WMR) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_kernel(const GemmArgs a) {
  static_assert(WMR == 2 || PB, "the half tile exists for packed weights only");
  constexpr int GMT = 128 * WMR;                                             // rows of this instantiation's tile
  constexpr int SLOTB = (PB && WMR == 1) ? GMT * GROWB : GSTAGEB;            // bytes of a ring slot (the half tile stages A only)
  constexpr int RINGN = (PB && WMR == 1) ? 4 : GRING;             // half tile: 8 KiB slots; eight of them (distance 7) measured no better than four
  constexpr int PD = RINGN - 1;                                              // A is requested PD half-stages ahead
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // Tile of this workgroup.  Workgroup i runs on XCD i % 8 and every XCD has its own L2; the ~32 workgroups an XCD holds at a time should
  // share operand panels.  The tiles are therefore walked in blocks of (32 / bc) row tiles x bc column tiles (bc = a.blk_c <= 4: 8 x 4 where
  // the grid allows, 8 + 4 panels fetched into that L2 for 32 tiles instead of 1 + 32), and XCD x owns a CONTIGUOUS range of that walk
  // (bijective split for any tile count: the first n % 8 XCDs take one tile more).  PMC / timing: 8192^3 908 -> 1 2xx TFLOP/s.
  int mt_i, nt_i;
  {
    const int n_tiles = a.n_mt * a.n_nt;
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + l;
    const int bc = a.blk_c, br = 32 / bc;
    const int band_tiles = br * a.n_nt;
    const int band = t / band_tiles, rem = t - band * band_tiles;
    const int rows = min(br, a.n_mt - band * br);            // the last band may be short
    const int blk = rem / (rows * bc), in = rem - blk * rows * bc;
    mt_i = band * br + in / bc;
    nt_i = blk * bc + in % bc;
  }
  const int m0 = mt_i * GMT, n0 = nt_i * GN;
  const int bz = blockIdx.y;
  const unsigned short* const xb = a.x + (size_t)bz * a.sx;

  const i32x4 ra = raw_rsrc(xb, 0x7fffffffu);
  const i32x4 rb = raw_rsrc(a.w + (size_t)bz * a.sw, 0x7fffffffu);
  // DMA geometry: one instruction = 16 rows x 64 bytes; lane -> (row lane >> 2, LDS slot lane & 3); the slot holds source chunk
  // slot ^ ((row >> 2) & 3).  Wave w fetches rows [32 w, 32 w + 32) of both tiles (two instructions each).
  const int lrow = lane >> 2, lslot = lane & 3;
  const int lchunk = lslot ^ ((4 - ((lrow >> 2) & 3)) & 3);
  int offa[2], offb[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = 32 * wave + 16 * q + lrow;
    const long long m = m0 + r < a.M ? m0 + r : a.M - 1;
    const long long n = n0 + r < a.N ? n0 + r : a.N - 1;
    offa[q] = (int)((m * a.lda + 8 * lchunk) * 2);
    offb[q] = (int)((n * a.ldw + 8 * lchunk) * 2);
  }
  // the DMAs of a half-stage in two halves (A rows, B rows): one half is issued in the LOAD phase, the other between the MFMAs of the
  // following MFMA phase -- all four in the LOAD phase made it longer than the partner's MFMA phase (measured), all four among the MFMAs
  // made that phase the longer one
#define KOFF(s) ((s) * GKH * 2)
  auto issue_a = [&](int s) {
    char* const st = smem + (s & (RINGN - 1)) * SLOTB;
#pragma unroll
    for (int q = 0; q < 2; ++q) lds_dma16(ra, st + (32 * wave + 16 * q) * GROWB, offa[q], KOFF(s));
  };
  auto issue_b = [&](int s, int q) {
    char* const st = smem + (s & (GRING - 1)) * GSTAGEB;
    lds_dma16(rb, st + GHALFB + (32 * wave + 16 * q) * GROWB, offb[q], KOFF(s));
  };
  auto issue = [&](int s) { issue_a(s); issue_b(s, 0); issue_b(s, 1); };
  // fragment read offsets: lane (r = lane & 31, h = lane >> 5) reads chunk (2 ks + h) ^ ((r >> 2) & 3) of its row
  // fragment reads of v_mfma_f32_16x16x32_bf16: lane (r = lane & 15, c = lane >> 4) reads chunk c of row r -- the whole 32-deep
  // half-stage is ONE k-step; the swizzle table {0, 3, 2, 1}[(row >> 2) & 3] makes every 16-lane group of a ds_read_b128 (lanes
  // {0-3, 12-15, 20-27}, ...) cover all 64 banks for this pattern
  const int fr = lane & 15, fc = lane >> 4;
  const int fsw = (4 - ((fr >> 2) & 3)) & 3;
  const int fa0 = (wm * 128 + fr) * GROWB + ((fc ^ fsw) << 4);
  const int fb0 = GHALFB + (wn * 64 + fr) * GROWB + ((fc ^ fsw) << 4);

  f32x4 acc[8][4];                                 // 16 x 16 blocks: rows 16 i .., columns 16 j ..
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int S = a.K / GKH;
  // Ping-pong between the two waves of a SIMD (wave w and w + 4: one of each wave row).  A wave alternates a LOAD phase -- read the 12
  // fragments of half-stage s, issue the DMAs of half-stage s + 3, retire its DMAs of s + 1 -- with an MFMA phase of 16 back-to-back
  // MFMAs; a barrier after every phase, and the second wave row runs ONE PHASE BEHIND the first, so on every SIMD one wave multiplies
  // while the other loads (the staggered form of cdna_hip_programming.md's 8-phase template).
  //   ring safety: L(s) overwrites the slot of half-stage s - 1, which both rows finished reading at least one barrier earlier;
  //   L(s) reads half-stage s, whose DMAs every wave retired (counted vmcnt) at the end of its own L(s - 1), a barrier earlier.
  s16x8 fa[8], fb[4];
  auto load_phase = [&](int s) {
    const char* const st = smem + (s & (GRING - 1)) * GSTAGEB;
    {
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const s16x8*>(st + fa0 + i * 16 * GROWB);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const s16x8*>(st + fb0 + j * 16 * GROWB);
    }
    if (s + GRING - 1 < S) issue_a(s + GRING - 1);
    // this wave's DMAs of half-stage s + 1 have landed; those of s + 2 and the A half of s + 3 (where they exist) may stay in flight
    // (the B half of s + 3 follows in the MFMA phase)
    const int after = S - 2 - s < 2 ? S - 2 - s : 2;
    if (after >= 2) vmw<6>(); else if (after == 1) vmw<4>(); else vmw<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // the fragment reads have returned: the slot may be refilled after the barrier
  };
  auto mfma_phase = [&](int s) {
    const bool more = s + GRING - 1 < S;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 4 * half; i < 4 * half + 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (more) issue_b(s + GRING - 1, half);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto phase_barrier = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  if constexpr (PB) {
    // ---- B fragments from global memory into a 3-deep register ring; A through the LDS ring as before (its B halves stay unused) ----
    const int S_ = S;
    const i32x4 rf = raw_rsrc(a.wf, 0x7fffffffu);
    int offf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int nb = (n0 + wn * 64) / 16 + j;
      nb = nb < a.N / 16 ? nb : a.N / 16 - 1;                               // column blocks beyond N: any valid fragment, results are masked
      offf[j] = nb * S_ * 1024 + lane * 16;
    }
    // the loads are inline asm: hipcc does not count the LDS-DMAs in its vmcnt model, so for loads it does track it would insert waits
    // that are too strict (measured: vmcnt(0) at every use, 2-3x slower); with asm loads every wait of the loop is one of the counted
    // vmw<> below, each followed by a sched_barrier so that no MFMA moves above it
    s16x8 fbr[2][4];
    auto load_b = [&](s16x8 (&slot)[4], int s) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(slot[j]) : "v"(offf[j]), "s"(rf), "s"(s * 1024) : "memory");
    };
    // issue order of a wave: A0 B0 A1 A2 | B1 A3 | B2 A4 | ...  (A = 2 LDS-DMAs, B = 4 loads into the register set the previous step's
    // MFMAs have just released).  At the end of LOAD(s) everything up to B(s) has to be there -- B(s) for this step's MFMAs, A(s+1),
    // issued before it, for the next step's reads a barrier later: A(s+2) B(s+1) A(s+3) may stay in flight, 8 operations, fewer at the end.
    auto vm_dyn = [&](int n) {                                                // n even
      switch (n) {
        case 0: vmw<0>(); break;  case 2: vmw<2>(); break;  case 4: vmw<4>(); break;  case 6: vmw<6>(); break;  case 8: vmw<8>(); break;
        case 10: vmw<10>(); break;  case 12: vmw<12>(); break;  case 14: vmw<14>(); break;  default: vmw<16>(); break;
      }
    };
    // (general prefetch distance PD: after B(s) a wave has issued A(s-1+PD), B(s+1), A(s+PD) -- where those half-stages exist)
    auto wait_tail = [&](int s) {
      const int left = S_ - 1 - s;                                            // steps after s
      if (left >= PD) vmw<8>(); else vm_dyn((left >= PD - 1 ? 2 : 0) + (left >= 1 ? 4 : 0));
    };
    issue_a(0);
    load_b(fbr[0], 0);
    for (int p = 1; p < PD && p < S_; ++p) issue_a(p);
    vm_dyn(4 + 2 * ((S_ < PD ? S_ : PD) - 1));                                // A(0) has landed
    phase_barrier();
    if (WMR == 2 && wm == 1) phase_barrier();
    auto step = [&](int s, s16x8 (&cur)[4], s16x8 (&nxt)[4]) {
      const char* const st = smem + (s & (RINGN - 1)) * SLOTB;
      if (s + 1 < S_) load_b(nxt, s + 1);
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const s16x8*>(st + fa0 + i * 16 * GROWB);
      if (s + PD < S_) issue_a(s + PD);
      wait_tail(s);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      phase_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], cur[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (WMR == 2) phase_barrier();   // one wave row: the barrier after LOAD already orders ring reads, DMA waits and refills
    };
    int s = 0;
    for (; s + 2 <= S_; s += 2) {
      step(s, fbr[0], fbr[1]);
      step(s + 1, fbr[1], fbr[0]);
    }
    if (s < S_) step(s, fbr[0], fbr[1]);
  } else {
  for (int s = 0; s < GRING - 1 && s < S; ++s) issue(s);
    if (S > 2) vmw<8>(); else if (S > 1) vmw<4>(); else vmw<0>();
    phase_barrier();                                                        // half-stage 0 is in LDS
    if (wm == 1) phase_barrier();                                           // the second wave row starts one phase later
    for (int s = 0; s < S; ++s) {
      load_phase(s);
      __builtin_amdgcn_sched_barrier(0);
      phase_barrier();
      mfma_phase(s);
      __builtin_amdgcn_sched_barrier(0);
      phase_barrier();
    }
  }
  if (WMR == 2 && wm == 0) phase_barrier();                               // the first row's partner of the extra barrier above
  // ---- epilogue ------------------------------------------------------------------------------------------------------
  // The accumulator layout (lane = column, registers = rows) would store 2 or 4 bytes per lane; instead each wave stages its tile, 32
  // columns at a time, as f32 [128 rows][32 columns] in its own 16 KiB of the (now idle) operand ring and reads it back row-wise:
  // 16 bytes per lane for the residual load and the f32 store, 8 bytes per lane for the bf16 store.
  const size_t yoff = (size_t)bz * a.sy;
  __builtin_amdgcn_s_barrier();                    // every wave is done with the operand ring (all DMAs were drained in the loop)
  asm volatile("" ::: "memory");
  const int erow = lane >> 3;
  if (!a.y && !a.res) {
    // bf16 result only (q / k / v, the first feed-forward linear, the conv layers): the wave's whole 128 x 64 tile is staged as bf16, row
    // pitch 144 bytes (the four row groups of a store instruction land 16 banks apart), and leaves as 128-byte row segments -- full
    // cache lines; 64-byte segments (32 columns at a time) cost 350 us of a 1 210 us 8192^3 product
    constexpr int EP16 = 144;
    char* const ep16 = smem + wave * (128 * EP16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nb = n0 + wn * 64 + 16 * j + fr;
      const float bv = (a.bias && nb < a.N) ? a.bias[nb] : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[i][j][r] + bv;
          if (a.act & 1) v = gelu_g(v);
          *reinterpret_cast<unsigned short*>(ep16 + (16 * i + 4 * fc + r) * EP16 + (16 * j + fr) * 2) = (unsigned short)pack_bf16(v, 0.f);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // wave-private tile: LDS operations of one wave are in order
    const int nb = n0 + wn * 64 + (lane & 7) * 8;
    if (nb < a.N) {
#pragma unroll 4
      for (int q = 0; q < 16; ++q) {
        const int row = 8 * q + erow;
        const int m = m0 + wm * 128 + row;
        if (m < a.M)
          *reinterpret_cast<u32x4*>(a.y16 + yoff + (size_t)m * a.ld16 + nb) = *reinterpret_cast<const u32x4*>(ep16 + row * EP16 + (lane & 7) * 16);
      }
    }
    return;
  }
  float* const ep = reinterpret_cast<float*>(smem + wave * 16384);
  const int ecol = (lane & 7) * 4;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int nb = n0 + wn * 64 + 32 * j;
    float bv[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) bv[jj] = (a.bias && nb + 16 * jj + fr < a.N) ? a.bias[nb + 16 * jj + fr] : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[i][2 * j + jj][r] + bv[jj];
          if (a.act & 1) v = gelu_g(v);
          ep[(16 * i + 4 * fc + r) * 32 + 16 * jj + fr] = v;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // wave-private tile: LDS operations of one wave are in order
    if (nb + ecol < a.N) {
#pragma unroll 4
      for (int q = 0; q < 16; ++q) {
        const int row = 8 * q + erow;
        const int m = m0 + wm * 128 + row;
        if (m < a.M) {
          f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * 32 + ecol);
          if (a.res) v += *reinterpret_cast<const f32x4*>(a.res + yoff + (size_t)m * a.ld_res + nb + ecol);
          if (a.y) *reinterpret_cast<f32x4*>(a.y + yoff + (size_t)m * a.ldc + nb + ecol) = v;
          if (a.y16) *reinterpret_cast<u32x2*>(a.y16 + yoff + (size_t)m * a.ld16 + nb + ecol) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// y = act(x w^T + bias) + res; see the header of this file.  TS_EUNSUPPORTED for shapes the kernel does not take.
// w as MFMA B fragments: out[N / 16][K / 32][64][8] bf16, lane (n = lane & 15, c = lane >> 4) of fragment (nb, s) holds
// w[16 nb + n][32 s + 8 c .. + 7] -- N * K elements, N % 16 == 0 and K % 32 == 0
__global__ void gemm_pack_w_kernel(const unsigned short* __restrict__ w, long long ldw, int N, int K, unsigned short* __restrict__ out) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte group
  const int S = K / GKH;
  if (g >= (long long)(N / 16) * S * 64) return;
  const int lane = (int)(g & 63);
  const long long f = g >> 6;
  const int s = (int)(f % S), nb = (int)(f / S);
  const unsigned short* src = w + (size_t)(16 * nb + (lane & 15)) * ldw + 32 * s + 8 * (lane >> 4);
  *reinterpret_cast<u32x4*>(out + g * 8) = *reinterpret_cast<const u32x4*>(src);
}

int gemm_nt_pack_w(hipStream_t stream, const void* w, long long ldw, int N, int K, void* out) {
  if (!w || !out || N <= 0 || K <= 0) return TS_EINVAL;
  if (N % 16 || K % GKH || ldw % 8 || ldw < K || (reinterpret_cast<uintptr_t>(w) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return TS_EUNSUPPORTED;
  const long long groups = (long long)(N / 16) * (K / GKH) * 64;
  (void)hipGetLastError();
  hipLaunchKernelGGL(gemm_pack_w_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, stream, static_cast<const unsigned short*>(w), ldw, N, K,
                     static_cast<unsigned short*>(out));
  return hip_status(hipGetLastError());
}

// wf: w in fragment order (gemm_nt_pack_w) or null
static int gemm_nt_bf16_sw(hipStream_t stream, const void* x, long long lda, long long sx, const void* w, long long ldw, const float* bias,
                           const float* res, long long ld_res, float* y, long long ldc, void* y16, long long ld16, long long sy, long long M, int N, int K,
                           int gelu, int batch, const void* wf, long long sw) {
  if (!x || !w || (!y && !y16) || M <= 0 || N <= 0 || K <= 0 || batch <= 0) return TS_EINVAL;
  if (N % 32 || K % GKH || lda % 8 || ldw % 8 || ldw < K) return TS_EUNSUPPORTED;
  if ((y && (ldc % 4 || (reinterpret_cast<uintptr_t>(y) & 15))) || (y16 && (ld16 % 8 || (reinterpret_cast<uintptr_t>(y16) & 15))) ||
      (res && (ld_res % 4 || (reinterpret_cast<uintptr_t>(res) & 15))) || sy % (y16 ? 8 : 4))
    return TS_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w) & 15) || (sx % 8)) return TS_EUNSUPPORTED;
  // 32-bit buffer offsets: the last row reads up to ((M - 1) lda + K) elements (overlapping conv rows have K > lda)
  if (((M - 1) * lda + K) * 2 >= (1ll << 31) - 1 || ((long long)(N - 1) * ldw + K) * 2 >= (1ll << 31) - 1) return TS_EUNSUPPORTED;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TS_EINVAL;
  static bool attr[64] = {};        // the attribute belongs to the (function, device) pair
  if (!attr[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS) != hipSuccess)
      return TS_EUNSUPPORTED;
    attr[dev] = true;
  }
  if (wf && (reinterpret_cast<uintptr_t>(wf) & 15)) return TS_EUNSUPPORTED;
  GemmArgs a;
  a.x = static_cast<const unsigned short*>(x); a.w = static_cast<const unsigned short*>(w); a.wf = static_cast<const unsigned short*>(wf); a.bias = bias; a.res = res; a.y = y;
  a.y16 = static_cast<unsigned short*>(y16);
  a.lda = lda; a.ldw = ldw; a.ld_res = ld_res; a.ldc = ldc; a.ld16 = ld16; a.sx = sx; a.sy = sy; a.sw = sw;
  a.M = (int)M; a.N = N; a.K = K; a.act = gelu ? 1 : 0;
  a.n_mt = (int)((M + GM - 1) / GM); a.n_nt = (N + GN - 1) / GN;
  a.blk_c = a.n_nt % 4 == 0 ? 4 : (a.n_nt % 3 == 0 ? 3 : (a.n_nt % 2 == 0 ? 2 : 1));
  (void)hipGetLastError();
  const dim3 grid((unsigned)(a.n_mt * a.n_nt), (unsigned)batch);
  if (wf) {                                        // packed weights: 128-row tiles, two workgroups per CU
    constexpr int LDS_H = 4 * 128 * 144;
    static bool attr_h[64] = {};
    if (!attr_h[dev]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_H) != hipSuccess)
        return TS_EUNSUPPORTED;
      attr_h[dev] = true;
    }
    a.n_mt = (int)((M + 127) / 128);
    hipLaunchKernelGGL((gemm_nt_kernel<true, 1>), dim3((unsigned)(a.n_mt * a.n_nt), (unsigned)batch), dim3(256), LDS_H, stream, a);
  } else hipLaunchKernelGGL(gemm_nt_kernel<false>, grid, dim3(512), GEMM_LDS, stream, a);
  return hip_status(hipGetLastError());
}

int gemm_nt_bf16(hipStream_t stream, const void* x, long long lda, long long sx, const void* w, long long ldw, const float* bias,
                 const float* res, long long ld_res, float* y, long long ldc, void* y16, long long ld16, long long sy, long long M, int N, int K,
                 int gelu, int batch, const void* wf) {
  return gemm_nt_bf16_sw(stream, x, lda, sx, w, ldw, bias, res, ld_res, y, ldc, y16, ld16, sy, M, N, K, gelu, batch, wf, 0);
}

}  // namespace ts

/* C-ABI form (tests, tools): y[m][n] = act(sum_k x[m][k] w[n][k] + bias[n]) + res[m][n]; see include/thunder_speech_amd.h */
extern "C" int ts_gemm_nt_bf16(const void* x, int64_t lda, const void* w, int64_t ldw, const float* bias, const float* res, int64_t ld_res,
                               float* y, int64_t ldc, void* y_bf16, int64_t ld16, int64_t rows, int32_t n, int32_t k, int32_t gelu, void* stream) {
  if (lda < 0 || ldw < k || (y && ldc < n) || (y_bf16 && ld16 < n) || (res && ld_res < n)) return TS_EINVAL;
  return ts::gemm_nt_bf16((hipStream_t)stream, x, lda, 0, w, ldw, bias, res, ld_res, y, ldc, y_bf16, ld16, 0, rows, n, k, gelu, 1, nullptr);
}

/* split-K form for products with few output tiles and a long contraction (the weight gradients of mixed-precision fine-tuning: 1024 x 1024 outputs
   = 16 tiles of 256 x 256 over 4 000 rows): `splits` launches-in-one (grid.y), split z multiplies columns [z k / splits, (z + 1) k / splits) of both
   operands into parts[z] (f32 [rows][n], pitch n); ts_w2v_sum_parts adds the parts in order. */
extern "C" int ts_gemm_nt_bf16_splitk(const void* x, int64_t lda, const void* w, int64_t ldw, float* parts, int64_t rows, int32_t n, int32_t k, int32_t splits,
                                      void* stream) {
  if (splits < 1 || k % splits || (k / splits) % 32 || lda < k || ldw < k) return TS_EINVAL;
  const long long ks = k / splits;
  if ((ks % 8) != 0) return TS_EUNSUPPORTED;
  return ts::gemm_nt_bf16_sw((hipStream_t)stream, x, lda, ks, w, ldw, nullptr, nullptr, 0, parts, n, nullptr, 0, (long long)rows * n, rows, n, (int)ks, 0, splits, nullptr, ks);
}

/* the same product with the weights ALSO given in fragment order (ts_gemm_nt_pack_w): the B operand then bypasses LDS */
extern "C" int ts_gemm_nt_pack_w(const void* w, int64_t ldw, int32_t n, int32_t k, void* w_frag, void* stream) {
  return ts::gemm_nt_pack_w((hipStream_t)stream, w, ldw, n, k, w_frag);
}
extern "C" int ts_gemm_nt_bf16_packed(const void* x, int64_t lda, const void* w, int64_t ldw, const void* w_frag, const float* bias, const float* res,
                                      int64_t ld_res, float* y, int64_t ldc, void* y_bf16, int64_t ld16, int64_t rows, int32_t n, int32_t k,
                                      int32_t gelu, void* stream) {
  if (!w_frag || lda < 0 || ldw < k || (y && ldc < n) || (y_bf16 && ld16 < n) || (res && ld_res < n)) return TS_EINVAL;
  return ts::gemm_nt_bf16((hipStream_t)stream, x, lda, 0, w, ldw, bias, res, ld_res, y, ldc, y_bf16, ld16, 0, rows, n, k, gelu, 1, w_frag);
}
