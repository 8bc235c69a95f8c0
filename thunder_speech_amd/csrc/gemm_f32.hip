// General f32-accumulating GEMM on the f32 matrix-core instruction (v_mfma_f32_32x32x2_f32: exact f32 products, 155 TFLOP/s on MI355X), for every
// product the library used to hand to rocBLAS: the "reference arithmetic" (f32) mode of the training path's 1x1 convolutions
// (quartznet/blocks.py:181 is an f32 conv1d), wav2vec2's precision="fp32" mode, and the rare bf16-operand products that have no kernel of their
// own (attention with a head size other than 64, grouped positional convs other than 64 channels, the decoder's f32 logits).
//
//   C[z][m][n] = sum_{j < nkb} sum_{k < K} A(z, j; m, k) B(z, j; k, n)  (+ C[z][m][n] if beta)  (+ bias[n])
//   A(m, k) at a + z sa + j ska + m a_rs + k a_cs,   B(k, n) at b + z sb + j skb + k b_rs + n b_cs  -- one of the two strides of each is 1
//
// so N/T forms and the "contraction runs over (clip, frame)" weight gradient are all the same kernel.  128 x 128 x 16 tiles, 4 waves as 2 x 2,
// wave tile 64 x 64 = 2 x 2 accumulators of 32 x 32; both operands are staged K-MAJOR in LDS ([k][row], pitch 132 words) whatever their layout in
// memory -- a K-contiguous operand is transposed by its four ds_write_b32 per 16-byte load, an M/N-contiguous one lands with one ds_write_b128 --
// so the fragment reads (one ds_read_b32 per operand and k-pair: lane l reads row l & 31 of k-row l >> 5) are the same conflict-free pattern
// for every form.  Next tile's global loads are issued before the current tile's products; 34 KiB of LDS and < 128 VGPRs leave room for four
// workgroups per CU, which is where the latency hiding comes from (no software pipeline beyond the register stage).
// bf16 operands are widened on the way into LDS (exact), the arithmetic is the same f32 instruction.
#include "ts_common.hpp"

namespace ts {

namespace {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LP = 132;                              // LDS row pitch in words: 16-byte aligned rows, k-rows 4 apart fall on banks 16 apart

struct GemmF32Args {
  const void* a; const void* b; void* c; const float* bias;
  long long a_rs, a_cs, b_rs, b_cs;
  long long ldc;
  long long sa, sb, sc;                              // batch strides: grid.z = z1 * nb2 + z2, operand offset z1 * s? + z2 * s?2
  long long sa2, sb2, sc2;
  int nb2;
  long long ska, skb;                                // strides of the outer contraction loop
  int M, N, K, nkb;
  int beta, out_bf16;
  int vec_a, vec_b;                                  // 1: 16-byte loads along the contiguous index (aligned base, pitches multiples of 4)
};

typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;

// 4 consecutive elements from `off`; vec: one 16-byte (bf16: 8-byte) load -- the caller guarantees alignment and that the row pitch covers the
// over-read; otherwise element by element, `nv` of them (the rest 0)
template <bool BF>
__device__ __forceinline__ f32x4 load4(const void* base, long long off, bool vec, int nv) {
  if (!vec) {
    f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < nv) r[i] = BF ? bf16_to_f32(static_cast<const unsigned short*>(base)[off + i]) : static_cast<const float*>(base)[off + i];
    return r;
  }
  if constexpr (BF) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(static_cast<const unsigned short*>(base) + off);
    return f32x4{bf16_lo(v[0]), bf16_hi(v[0]), bf16_lo(v[1]), bf16_hi(v[1])};
  } else {
    return *reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + off);
  }
}

// One operand tile: `rows` = M (or N) extent, contraction extent K; KC: the contraction index is the contiguous one in memory.
// Each thread fetches two 4-element vectors per stage.  The thread's element offsets at (j = 0, k0 = 0) are computed ONCE (`off`); rows beyond
// the extent are redirected to row 0 -- what they fetch only ever reaches output rows / columns that are not stored -- so the main loop's fetch
// is two unconditional loads and an add; only a stage that crosses K (or an operand without vector alignment) takes the guarded path.
template <bool KC, bool BF>
struct Stage {
  f32x4 v[2];
  long long off[2];
  long long step;                                    // elements per stage (BK along the contraction index)
  int kt[2], rr[2];                                  // this thread's k (relative to k0) and row of each vector (guarded path)
  __device__ __forceinline__ void init(long long rs, long long cs, int rows, int r0, int tid) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if constexpr (KC) {
        rr[p] = r0 + (tid >> 2) + 64 * p; kt[p] = 4 * (tid & 3);                     // 4 lanes cover the 16 k of a row
        off[p] = (long long)(rr[p] < rows ? rr[p] : 0) * rs + kt[p];
      } else {
        kt[p] = (tid >> 5) + 8 * p; rr[p] = r0 + 4 * (tid & 31);                     // 32 lanes cover the 128 rows of a k
        off[p] = (long long)kt[p] * cs + (rr[p] < rows ? rr[p] : 0);
      }
    }
    step = KC ? BK : BK * cs;
  }
  __device__ __forceinline__ void fetch_fast(const void* base, long long o) {
#pragma unroll
    for (int p = 0; p < 2; ++p) v[p] = load4<BF>(base, off[p] + o, true, 4);
  }
  // guarded: elements outside [0, rows) x [0, K) come back as 0
  __device__ __forceinline__ void fetch_slow(const void* base, long long o, int rows, int K, int k0, bool vec) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int k = k0 + kt[p];
      v[p] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (rr[p] < rows && k < K) {
        const int nv = KC ? K - k : rows - rr[p];
        v[p] = load4<BF>(base, off[p] + o, vec && nv >= 4, nv);     // a vector load never reaches past the operand's logical extent here
#pragma unroll
        for (int i = 1; i < 4; ++i) if (i >= nv) v[p][i] = 0.f;
      }
    }
  }
  __device__ __forceinline__ void store(float* tile, int tid) const {                // tile: [BK][LP]
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if constexpr (KC) {
        const int r = (tid >> 2) + 64 * p, k = 4 * (tid & 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) tile[(k + i) * LP + r] = v[p][i];
      } else {
        const int k = (tid >> 5) + 8 * p, r = 4 * (tid & 31);
        *reinterpret_cast<f32x4*>(tile + k * LP + r) = v[p];
      }
    }
  }
};

template <bool A_KC, bool B_KC, bool BF>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32Args g) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][BK * LP];               // [buffer][A | B][k][row]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n_nt = (g.N + BN - 1) / BN;
  const int m0 = (blockIdx.x / n_nt) * BM, n0 = (blockIdx.x % n_nt) * BN;
  const long long z1 = blockIdx.z / g.nb2, z2 = blockIdx.z % g.nb2;
  const char* const a0 = static_cast<const char*>(g.a) + (z1 * g.sa + z2 * g.sa2) * (BF ? 2 : 4);
  const char* const b0 = static_cast<const char*>(g.b) + (z1 * g.sb + z2 * g.sb2) * (BF ? 2 : 4);
  const int nks = (g.K + BK - 1) / BK, total = nks * g.nkb;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Stage<A_KC, BF> sa;
  Stage<B_KC, BF> sb;
  sa.init(g.a_rs, g.a_cs, g.M, m0, tid);
  // B(k, n): the "rows" of the staging helper are n; KC: k contiguous (b_rs == 1), the row stride is b_cs
  sb.init(B_KC ? g.b_cs : 0, B_KC ? 0 : g.b_rs, g.N, n0, tid);
  int sj = 0, sk = 0;                                 // (outer contraction step, stage inside it) of the next fetch
  auto fetch = [&]() {
    const int k0 = sk * BK;
    const long long oa = sj * g.ska + sk * sa.step, ob = sj * g.skb + sk * sb.step;
    const bool full = k0 + BK <= g.K;                 // wave-uniform; the fast path is chosen PER OPERAND: one unaligned operand (e.g. attention
    if (g.vec_a && full) sa.fetch_fast(a0, oa);       // probabilities with an odd frame count as their pitch) does not drag the other one down
    else sa.fetch_slow(a0, oa, g.M, g.K, k0, g.vec_a != 0);
    if (g.vec_b && full) sb.fetch_fast(b0, ob);
    else sb.fetch_slow(b0, ob, g.N, g.K, k0, g.vec_b != 0);
    if (++sk == nks) { sk = 0; ++sj; }
  };
  fetch();
  sa.store(lds[0][0], tid);
  sb.store(lds[0][1], tid);
  __syncthreads();
  const int fa = (lane >> 5) * LP + 64 * wm + (lane & 31), fb = (lane >> 5) * LP + 64 * wn + (lane & 31);
  for (int s = 0; s < total; ++s) {
    const int cur = s & 1;
    if (s + 1 < total) fetch();                                                     // in flight under this tile's products
    const float* const ta = lds[cur][0];
    const float* const tb = lds[cur][1];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const float a_0 = ta[fa + 2 * kk * LP], a_1 = ta[fa + 2 * kk * LP + 32];
      const float b_0 = tb[fb + 2 * kk * LP], b_1 = tb[fb + 2 * kk * LP + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_0, b_0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_0, b_1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_1, b_0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_1, b_1, acc[1][1], 0, 0, 0);
    }
    if (s + 1 < total) {
      sa.store(lds[cur ^ 1][0], tid);                                               // the other buffer: last read before the previous barrier
      sb.store(lds[cur ^ 1][1], tid);
    }
    __syncthreads();
  }
  // epilogue: accumulator (i, j), register r: row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
  char* const c0 = static_cast<char*>(g.c) + (z1 * g.sc + z2 * g.sc2) * (g.out_bf16 ? 2 : 4);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + 64 * wn + 32 * j + (lane & 31);
      if (n >= g.N) continue;
      const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= g.M) continue;
        const long long o = (long long)m * g.ldc + n;
        float v = acc[i][j][r] + bv;
        if (g.out_bf16) {
          unsigned short* const cp = reinterpret_cast<unsigned short*>(c0) + o;
          if (g.beta) v += bf16_to_f32(*cp);
          *cp = (unsigned short)(pack_bf16(v, 0.f) & 0xffffu);
        } else {
          float* const cp = reinterpret_cast<float*>(c0) + o;
          if (g.beta) v += *cp;
          *cp = v;
        }
      }
    }
}

}  // namespace

// Strides in ELEMENTS; exactly one of (a_rs, a_cs) and one of (b_rs, b_cs) must be 1 (TS_EUNSUPPORTED otherwise).  An operand whose base is 16-byte
// aligned (8 for bf16) and whose other strides are all multiples of 4 is fetched with vector loads -- a load may then reach 3 elements past the
// logical extent of an M/N-contiguous operand's row, which stays inside the row pitch (a multiple of 4) and only feeds outputs that are not
// stored; along K the last partial stage is fetched element by element -- any other operand element by element throughout.
// Two batch levels: grid.z = batch x batch2, operand offsets z1 s? + z2 s?2 ((clip, head) of the attention products; (group, tap) of the
// positional conv's weight gradient).
int gemm_f32_b2(hipStream_t stream, bool in_bf16, const void* a, long long a_rs, long long a_cs, long long sa, long long ska, const void* b,
                long long b_rs, long long b_cs, long long sb, long long skb, void* c, long long ldc, long long sc, bool out_bf16, const float* bias,
                int M, int N, int K, int nkb, int batch, bool beta, int batch2, long long sa2, long long sb2, long long sc2) {
  if (!a || !b || !c || M <= 0 || N <= 0 || K <= 0 || nkb <= 0 || batch <= 0 || batch2 <= 0 || ldc < N) return TS_EINVAL;
  if ((long long)batch * batch2 > 65535) return TS_EUNSUPPORTED;
  const bool a_kc = a_cs == 1, b_kc = b_rs == 1;
  if ((!a_kc && a_rs != 1) || (!b_kc && b_cs != 1)) return TS_EUNSUPPORTED;
  const long long a_ld = a_kc ? a_rs : a_cs, b_ld = b_kc ? b_cs : b_rs;
  const uintptr_t al = in_bf16 ? 7 : 15;
  const bool vec_a = !(a_ld % 4 || sa % 4 || sa2 % 4 || ska % 4 || (reinterpret_cast<uintptr_t>(a) & al));
  const bool vec_b = !(b_ld % 4 || sb % 4 || sb2 % 4 || skb % 4 || (reinterpret_cast<uintptr_t>(b) & al));
  GemmF32Args g{};
  g.a = a; g.b = b; g.c = c; g.bias = bias;
  g.a_rs = a_rs; g.a_cs = a_cs; g.b_rs = b_rs; g.b_cs = b_cs; g.ldc = ldc;
  g.sa = sa; g.sb = sb; g.sc = sc; g.ska = ska; g.skb = skb;
  g.sa2 = sa2; g.sb2 = sb2; g.sc2 = sc2; g.nb2 = batch2;
  g.M = M; g.N = N; g.K = K; g.nkb = nkb; g.beta = beta ? 1 : 0; g.out_bf16 = out_bf16 ? 1 : 0;
  g.vec_a = vec_a ? 1 : 0; g.vec_b = vec_b ? 1 : 0;
  const dim3 grid((unsigned)(((M + BM - 1) / BM) * ((N + BN - 1) / BN)), 1, (unsigned)(batch * batch2));
  (void)hipGetLastError();
#define TS_GF(AK, BK_, BF_) hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, BF_>), grid, dim3(256), 0, stream, g)
  if (in_bf16) {
    if (a_kc && b_kc) TS_GF(true, true, true); else if (a_kc) TS_GF(true, false, true); else if (b_kc) TS_GF(false, true, true); else TS_GF(false, false, true);
  } else {
    if (a_kc && b_kc) TS_GF(true, true, false); else if (a_kc) TS_GF(true, false, false); else if (b_kc) TS_GF(false, true, false); else TS_GF(false, false, false);
  }
#undef TS_GF
  return hip_status(hipGetLastError());
}

int gemm_f32(hipStream_t stream, bool in_bf16, const void* a, long long a_rs, long long a_cs, long long sa, long long ska, const void* b,
             long long b_rs, long long b_cs, long long sb, long long skb, void* c, long long ldc, long long sc, bool out_bf16, const float* bias,
             int M, int N, int K, int nkb, int batch, bool beta) {
  return gemm_f32_b2(stream, in_bf16, a, a_rs, a_cs, sa, ska, b, b_rs, b_cs, sb, skb, c, ldc, sc, out_bf16, bias, M, N, K, nkb, batch, beta, 1, 0, 0, 0);
}

}  // namespace ts

/* C-ABI form (tests, tools); see include/thunder_speech_amd.h */
extern "C" int ts_gemm_f32(const void* a, int64_t a_rs, int64_t a_cs, int64_t sa, int64_t ska, const void* b, int64_t b_rs, int64_t b_cs, int64_t sb,
                           int64_t skb, void* c, int64_t ldc, int64_t sc, const float* bias, int32_t m, int32_t n, int32_t k, int32_t nkb,
                           int32_t batch, int32_t in_bf16, int32_t out_bf16, int32_t beta, void* stream) {
  return ts::gemm_f32(reinterpret_cast<hipStream_t>(stream), in_bf16 != 0, a, a_rs, a_cs, sa, ska, b, b_rs, b_cs, sb, skb, c, ldc, sc, out_bf16 != 0,
                      bias, m, n, k, nkb, batch, beta != 0);
}

/* the same with a second batch level: grid.z = batch x batch2, operand offsets z1 s? + z2 s?2 */
extern "C" int ts_gemm_f32_b2(const void* a, int64_t a_rs, int64_t a_cs, int64_t sa, int64_t sa2, int64_t ska, const void* b, int64_t b_rs, int64_t b_cs,
                              int64_t sb, int64_t sb2, int64_t skb, void* c, int64_t ldc, int64_t sc, int64_t sc2, const float* bias, int32_t m,
                              int32_t n, int32_t k, int32_t nkb, int32_t batch, int32_t batch2, int32_t beta, void* stream) {
  return ts::gemm_f32_b2(reinterpret_cast<hipStream_t>(stream), false, a, a_rs, a_cs, sa, ska, b, b_rs, b_cs, sb, skb, c, ldc, sc, false, bias, m, n, k,
                         nkb, batch, beta != 0, batch2, sa2, sb2, sc2);
}
