// Fused self-attention for mixed-precision wav2vec2 FINE-TUNING (head_dim 64): forward and backward without the [T][T] score / probability
// matrices of the unfused path (huggingface/train.py Attention: 127 MB of f32 per layer at 8 x 10 s, written and re-read by six products, two
// softmax passes and two dropout passes -- 28 of the step's 58 ms, profiles/round6_c5_finetune.md).  The reference reaches this code through
// transformers' Wav2Vec2Attention inside AutoModelForCTC (thunder huggingface/compatibility.py:31-42) under Lightning's training_step
// (module.py:102-127): ctx = dropout(softmax(q k^T / sqrt(hd) + key mask)) v.
//
//   forward   (attn_fwd_train_kernel)  the inference kernel of csrc/w2v_enc.hip (S^T = K Q^T on v_mfma_f32_32x32x16_bf16, online softmax with a lane
//             owning one query column, O^T += V^T P^T with P^T straight out of the accumulators) + the dropout mask of ts_train_dropout drawn in the
//             kernel (Philox4x32-10, element e = ((b H + h) T + q) T + k -> word e & 3 of block e >> 2) + the row statistic lse2 = max + log2(sum)
//             (log2 domain, scale folded in) that lets the backward rebuild any probability as exp2(s c - lse2);
//   backward  two kernels, each recomputing the probabilities and the mask (no atomics, fixed summation order):
//             attn_bwd_dq_kernel    workgroup = 128 queries, loop over key tiles:   dQ^T[d][q] += K^T dS^T   (the forward's second product with K, dS)
//             attn_bwd_dkv_kernel   workgroup = 64 keys, loop over query tiles:     dV += Pd^T dO,  dK += dS^T Q   (P / dS tiles through wave-private LDS)
//             with dP = (dO V^T) * keep / (1 - p),  dS = P * (dP - D),  D[q] = sum_d dO[q][d] O[q][d]  (attn_rowdot_kernel, which also casts dO).
// Operands bf16 (q, k, v, dO, P, dS), accumulation and softmax arithmetic f32, results f32 (ctx, dqkv).
#include "ts_common.hpp"
#include "ts_philox.hpp"

namespace ts {

namespace {

constexpr int TA_KT = 64;          // keys per staged tile
constexpr int TA_PITCH = 144;      // bytes per staged row: 64 bf16 + 16
constexpr int TA_QW = 128;         // queries per workgroup (4 waves x 32)

struct TaArgs {
  const unsigned short* qkv;       // [B][T][3C] bf16
  const int* key_len;
  float* ctx;                      // forward: [B][T][C] f32
  float* lse2;                     // [B][H][T]
  const unsigned short* dout;      // backward: [B][T][C] bf16
  const float* dsum;               // backward: D [B][H][T]
  float* dqkv;                     // backward: [B][T][3C] f32
  int t, c, heads;
  float scale_log2e, scale;        // log2(e) / sqrt(hd), 1 / sqrt(hd)
  float p_drop, keep_scale;        // dropout probability, 1 / (1 - p)
  unsigned long long seed;
  const unsigned* mask;            // keep bits of the whole [B H T][T] dropout stream, bit e of the flat bitstring (attn_mask_kernel); NULL when p_drop == 0
};

// The dropout mask as a bitstring, drawn ONCE per call by its own small kernel: bit e is set iff element e of the logical [B * heads * T][T] probability
// matrix keeps its value -- ts_train_dropout's rule: u01(word e & 3 of Philox block e >> 2) >= p.  A thread draws the 8 blocks of one 32-bit word.  Drawing
// the mask inside the attention kernels cost three Philox blocks per 8 probabilities (T is odd: a lane's run of 8 keys straddles three blocks) in each of the
// three kernels -- more than the attention arithmetic itself; now they read two words per run.
__global__ __launch_bounds__(256) void attn_mask_kernel(unsigned* __restrict__ mask, long long n_words, unsigned long long seed, float p) {
  const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
  if (w >= n_words) return;
  unsigned bits = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const Philox4 r = philox(seed, PHILOX_DROPOUT, (unsigned long long)w * 8 + j);
#pragma unroll
    for (int q = 0; q < 4; ++q) bits |= (u01(r.v[q]) >= p ? 1u : 0u) << (4 * j + q);
  }
  mask[w] = bits;
}

// keep bits of the 8 consecutive elements e0 .. e0 + 7 (bit j: element e0 + j is kept)
__device__ __forceinline__ unsigned keep8(const unsigned* __restrict__ mask, unsigned long long e0) {
  const unsigned long long w = e0 >> 5;
  const unsigned long long both = ((unsigned long long)mask[w + 1] << 32) | mask[w];
  return (unsigned)(both >> (e0 & 31)) & 0xffu;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_fwd_train_kernel(const TaArgs a) {
  __shared__ __attribute__((aligned(16))) char ks_[TA_KT * TA_PITCH];
  __shared__ __attribute__((aligned(16))) char vs_[TA_KT * TA_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, head = blockIdx.y;
  const int q0 = blockIdx.x * TA_QW + wave * 32;
  const size_t rowp = (size_t)3 * a.c;
  const unsigned short* base = a.qkv + (size_t)b * a.t * rowp + (size_t)head * 64;
  int lim = a.t;
  if (a.key_len) {
    const int n = a.key_len[b] < a.t ? a.key_len[b] : a.t;
    lim = n > 0 ? n : 0;                         // no valid key: the training path's convention (ts_w2v_softmax_fwd) -- every probability 0, ctx = 0
  }
  const int half = lane >> 5, n32 = lane & 31;
  const int query = q0 + n32;
  const bool drop = a.p_drop > 0.f;
  const unsigned long long erow = ((unsigned long long)(b * a.heads + head) * a.t + (query < a.t ? query : a.t - 1)) * (unsigned long long)a.t;
  s16x8 qf[4];
  {
    const int qrow = query < a.t ? query : a.t - 1;
    const uint4* qp = reinterpret_cast<const uint4*>(base + (size_t)qrow * rowp + 8 * half);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = __builtin_bit_cast(s16x8, qp[2 * ks]);
  }
  f32x16 o[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[mt][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int pm = (n32 & ~12) | ((n32 & 4) << 1) | ((n32 & 8) >> 1);          // K row order: bits 2 and 3 swapped
  const int q4 = (lane >> 2) & 3, gq = (lane >> 4) & 1, p4 = lane & 3;
  const int v_off = (8 * half + q4) * TA_PITCH + (16 * gq + 4 * p4) * 2;     // transposing read of the V tile

  for (int k0 = 0; k0 < lim; k0 += TA_KT) {
    __syncthreads();
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int chunk = tid + 256 * rep, r = chunk >> 3, cc = chunk & 7;
      const int key = k0 + r < a.t ? k0 + r : a.t - 1;
      const unsigned short* src = base + (size_t)key * rowp + cc * 8;
      *reinterpret_cast<uint4*>(ks_ + r * TA_PITCH + cc * 16) = *reinterpret_cast<const uint4*>(src + a.c);
      *reinterpret_cast<uint4*>(vs_ + r * TA_PITCH + cc * 16) = *reinterpret_cast<const uint4*>(src + 2 * a.c);
    }
    __syncthreads();
    const bool full = k0 + TA_KT <= lim;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (k0 + sub * 32 >= lim) break;
      f32x16 s;
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = 0.f;
      const char* kr = ks_ + (sub * 32 + pm) * TA_PITCH + half * 16;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const s16x8*>(kr + ks * 32), qf[ks], s, 0, 0, 0);
      // accumulator register i <-> key kbase + 16 (i / 8) + i % 8
      const int kbase = k0 + sub * 32 + 8 * half;
      if (!full) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (kbase + 16 * (i >> 3) + (i & 7) >= lim) s[i] = -INFINITY;
      }
      float mx = s[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx * a.scale_log2e);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float rs = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = __builtin_amdgcn_exp2f(fmaf(s[i], a.scale_log2e, -m_new)); rs += s[i]; }
      l_run = l_run * alpha + rs;                                             // the normaliser counts every key, dropped or not
      m_run = m_new;
      if (drop) {
        const unsigned k_lo = keep8(a.mask, erow + kbase), k_hi = keep8(a.mask, erow + kbase + 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          s[i] = (k_lo >> i) & 1u ? s[i] * a.keep_scale : 0.f;
          s[8 + i] = (k_hi >> i) & 1u ? s[8 + i] * a.keep_scale : 0.f;
        }
      }
      if (__any(alpha != 1.f)) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[mt][i] *= alpha;
      }
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        const unsigned p01 = pack_bf16(s[8 * ks2 + 0], s[8 * ks2 + 1]), p23 = pack_bf16(s[8 * ks2 + 2], s[8 * ks2 + 3]);
        const unsigned p45 = pack_bf16(s[8 * ks2 + 4], s[8 * ks2 + 5]), p67 = pack_bf16(s[8 * ks2 + 6], s[8 * ks2 + 7]);
        const s16x8 pb = __builtin_bit_cast(s16x8, uint4{p01, p23, p45, p67});
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const char* va = vs_ + (sub * 32 + 16 * ks2) * TA_PITCH + v_off + 64 * mt;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)va));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)va + 4 * TA_PITCH));
          const s16x8 vf = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, o[mt], 0, 0, 0);
        }
      }
    }
  }
  const float l = l_run + __shfl_xor(l_run, 32);
  const float inv = l > 0.f ? 1.f / l : 0.f;     // (no valid key: zeros, and a row statistic that makes every rebuilt probability 0)
  if (query < a.t) {
    float* dst = a.ctx + ((size_t)b * a.t + query) * a.c + (size_t)head * 64 + 4 * half;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g)       // accumulator registers 4g .. 4g+3 <-> d = 32 mt + 8 g + 4 half + 0..3
        *reinterpret_cast<f32x4*>(dst + 32 * mt + 8 * g) = f32x4{o[mt][4 * g] * inv, o[mt][4 * g + 1] * inv, o[mt][4 * g + 2] * inv, o[mt][4 * g + 3] * inv};
    if (half == 0) a.lse2[((size_t)b * a.heads + head) * a.t + query] = l > 0.f ? m_run + __builtin_amdgcn_logf(l) : INFINITY;      // v_log_f32 = log2
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// backward, part 0: bf16 copy of dO and D[b][h][q] = sum_d dO[q][d] O[q][d] over the head's 64 columns.  One wave per (b, q) row, four columns per lane.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_rowdot_kernel(const float* __restrict__ dout, const float* __restrict__ ctx, unsigned short* __restrict__ dout16,
                                                          float* __restrict__ dsum, long long rows, int t, int c, int heads) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const long long b = row / t;
  const int q = (int)(row - b * t);
  for (int col = lane * 4; col < c; col += 256) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(dout + row * c + col), o = *reinterpret_cast<const f32x4*>(ctx + row * c + col);
    *reinterpret_cast<u32x2*>(dout16 + row * c + col) = u32x2{pack_bf16(g[0], g[1]), pack_bf16(g[2], g[3])};
    float d = (g[0] * o[0] + g[1] * o[1]) + (g[2] * o[2] + g[3] * o[3]);
    d += __shfl_xor(d, 8); d += __shfl_xor(d, 4); d += __shfl_xor(d, 2); d += __shfl_xor(d, 1);
    if ((lane & 15) == 0) dsum[((size_t)b * heads + col / 64) * t + q] = d;
  }
}

// probabilities, mask and dS of one 32-key sub-tile; registers i <-> key kbase + 16 (i / 8) + i % 8, lane <-> one query
//   in:  s = q . k (raw), dp = dO . v (raw);  out: s = P * keep / (1 - p) (the dV operand), dp = dS = P * (dP - D)
__device__ __forceinline__ void attn_bwd_tile(f32x16& s, f32x16& dp, int kbase, int lim, bool q_ok, float lse2, float dsum, const TaArgs& a, unsigned long long erow,
                                              bool drop) {
  unsigned k_lo = 0xffu, k_hi = 0xffu;
  if (drop) { k_lo = keep8(a.mask, erow + kbase); k_hi = keep8(a.mask, erow + kbase + 16); }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int key = kbase + 16 * (i >> 3) + (i & 7);
    float p = (q_ok && key < lim) ? __builtin_amdgcn_exp2f(fmaf(s[i], a.scale_log2e, -lse2)) : 0.f;
    const bool kept = (((i < 8 ? k_lo : k_hi) >> (i & 7)) & 1u) != 0;
    const float ks = kept ? a.keep_scale : 0.f;
    const float dpv = dp[i] * ks;
    dp[i] = p * (dpv - dsum);
    s[i] = p * ks;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, dQ: workgroup = 128 queries of one (clip, head); K / V tiles of 64 keys through LDS; dQ^T[d][q] += K^T dS^T with dS^T out of the accumulators
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const TaArgs a) {
  __shared__ __attribute__((aligned(16))) char ks_[TA_KT * TA_PITCH];
  __shared__ __attribute__((aligned(16))) char vs_[TA_KT * TA_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, head = blockIdx.y;
  const int q0 = blockIdx.x * TA_QW + wave * 32;
  const size_t rowp = (size_t)3 * a.c;
  const unsigned short* base = a.qkv + (size_t)b * a.t * rowp + (size_t)head * 64;
  int lim = a.t;
  if (a.key_len) {
    const int n = a.key_len[b] < a.t ? a.key_len[b] : a.t;
    lim = n > 0 ? n : 0;
  }
  const int half = lane >> 5, n32 = lane & 31;
  const int query = q0 + n32;
  const bool q_ok = query < a.t;
  const int qrow = q_ok ? query : a.t - 1;
  const bool drop = a.p_drop > 0.f;
  const unsigned long long erow = ((unsigned long long)(b * a.heads + head) * a.t + qrow) * (unsigned long long)a.t;
  s16x8 qf[4], gf[4];
  {
    const uint4* qp = reinterpret_cast<const uint4*>(base + (size_t)qrow * rowp + 8 * half);
    const uint4* gp = reinterpret_cast<const uint4*>(a.dout + ((size_t)b * a.t + qrow) * a.c + (size_t)head * 64 + 8 * half);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { qf[ks] = __builtin_bit_cast(s16x8, qp[2 * ks]); gf[ks] = __builtin_bit_cast(s16x8, gp[2 * ks]); }
  }
  const float lse2 = a.lse2[((size_t)b * a.heads + head) * a.t + qrow], dsum = a.dsum[((size_t)b * a.heads + head) * a.t + qrow];
  f32x16 dq[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[mt][i] = 0.f;
  const int pm = (n32 & ~12) | ((n32 & 4) << 1) | ((n32 & 8) >> 1);
  const int q4 = (lane >> 2) & 3, gq = (lane >> 4) & 1, p4 = lane & 3;
  const int v_off = (8 * half + q4) * TA_PITCH + (16 * gq + 4 * p4) * 2;
  for (int k0 = 0; k0 < lim; k0 += TA_KT) {
    __syncthreads();
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int chunk = tid + 256 * rep, r = chunk >> 3, cc = chunk & 7;
      const int key = k0 + r < a.t ? k0 + r : a.t - 1;
      const unsigned short* src = base + (size_t)key * rowp + cc * 8;
      *reinterpret_cast<uint4*>(ks_ + r * TA_PITCH + cc * 16) = *reinterpret_cast<const uint4*>(src + a.c);
      *reinterpret_cast<uint4*>(vs_ + r * TA_PITCH + cc * 16) = *reinterpret_cast<const uint4*>(src + 2 * a.c);
    }
    __syncthreads();
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (k0 + sub * 32 >= lim) break;
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
      const char* kr = ks_ + (sub * 32 + pm) * TA_PITCH + half * 16;
      const char* vr = vs_ + (sub * 32 + pm) * TA_PITCH + half * 16;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const s16x8*>(kr + ks * 32), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const s16x8*>(vr + ks * 32), gf[ks], dp, 0, 0, 0);
      }
      attn_bwd_tile(s, dp, k0 + sub * 32 + 8 * half, lim, q_ok, lse2, dsum, a, erow, drop);
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        const unsigned p01 = pack_bf16(dp[8 * ks2 + 0], dp[8 * ks2 + 1]), p23 = pack_bf16(dp[8 * ks2 + 2], dp[8 * ks2 + 3]);
        const unsigned p45 = pack_bf16(dp[8 * ks2 + 4], dp[8 * ks2 + 5]), p67 = pack_bf16(dp[8 * ks2 + 6], dp[8 * ks2 + 7]);
        const s16x8 pb = __builtin_bit_cast(s16x8, uint4{p01, p23, p45, p67});
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const char* ka = ks_ + (sub * 32 + 16 * ks2) * TA_PITCH + v_off + 64 * mt;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)ka));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)ka + 4 * TA_PITCH));
          const s16x8 kf = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          dq[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, pb, dq[mt], 0, 0, 0);
        }
      }
    }
  }
  if (q_ok) {
    float* dst = a.dqkv + ((size_t)b * a.t + query) * rowp + (size_t)head * 64 + 4 * half;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(dst + 32 * mt + 8 * g) =
            f32x4{dq[mt][4 * g] * a.scale, dq[mt][4 * g + 1] * a.scale, dq[mt][4 * g + 2] * a.scale, dq[mt][4 * g + 3] * a.scale};
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, dK and dV: workgroup = 64 keys of one (clip, head), loop over 128-query tiles (32 per wave); per wave and tile the Q and dO rows are staged
// in wave-private LDS, P * keep / (1 - p) and dS go there as bf16 [query][key] tiles, and
//   dV^T[d][key] += dO^T[d][q] Pd[q][key],   dK^T[d][key] += Q^T[d][q] dS[q][key]
// take both operands out of those tiles with transposing reads (contraction over the tile's 32 queries); the four waves' sums meet in LDS at the end.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TA_WTILE = 32 * TA_PITCH;       // one [32 rows][64 columns] bf16 tile

__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const TaArgs a) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* const ks_ = sm;                                  // [64][PITCH]
  char* const vs_ = sm + TA_KT * TA_PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* const qt = sm + 2 * TA_KT * TA_PITCH + wave * 4 * TA_WTILE;     // this wave's Q, dO, Pd, dS tiles
  char* const gt = qt + TA_WTILE;
  char* const pt = gt + TA_WTILE;
  char* const st = pt + TA_WTILE;
  const int b = blockIdx.z, head = blockIdx.y, k0 = blockIdx.x * TA_KT;
  const size_t rowp = (size_t)3 * a.c;
  const unsigned short* base = a.qkv + (size_t)b * a.t * rowp + (size_t)head * 64;
  int lim = a.t;
  if (a.key_len) {
    const int n = a.key_len[b] < a.t ? a.key_len[b] : a.t;
    lim = n > 0 ? n : 0;
  }
  const bool drop = a.p_drop > 0.f;
  const int half = lane >> 5, n32 = lane & 31;
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    const int chunk = tid + 256 * rep, r = chunk >> 3, cc = chunk & 7;
    const int key = k0 + r < a.t ? k0 + r : a.t - 1;
    const unsigned short* src = base + (size_t)key * rowp + cc * 8;
    *reinterpret_cast<uint4*>(ks_ + r * TA_PITCH + cc * 16) = *reinterpret_cast<const uint4*>(src + a.c);
    *reinterpret_cast<uint4*>(vs_ + r * TA_PITCH + cc * 16) = *reinterpret_cast<const uint4*>(src + 2 * a.c);
  }
  __syncthreads();
  f32x16 dv[2][2], dk[2][2];                             // [d block mt][key block nt]
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) { dv[mt][nt][i] = 0.f; dk[mt][nt][i] = 0.f; }
  const int pm = (n32 & ~12) | ((n32 & 4) << 1) | ((n32 & 8) >> 1);
  const int q4 = (lane >> 2) & 3, gq = (lane >> 4) & 1, p4 = lane & 3;
  const int tr_off = (8 * half + q4) * TA_PITCH + (16 * gq + 4 * p4) * 2;       // transposing read: rows = contraction index, columns = M / N index
  const bool any_key = k0 < lim;
  for (int q0 = wave * 32; q0 < a.t && any_key; q0 += TA_QW) {
    // ---- stage this wave's 32 query rows of Q and dO (wave-private: LDS operations of a wave execute in order)
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      const int chunk = lane + 64 * rep, r = chunk >> 3, cc = chunk & 7;
      const int qr = q0 + r < a.t ? q0 + r : a.t - 1;
      *reinterpret_cast<uint4*>(qt + r * TA_PITCH + cc * 16) = *reinterpret_cast<const uint4*>(base + (size_t)qr * rowp + cc * 8);
      *reinterpret_cast<uint4*>(gt + r * TA_PITCH + cc * 16) = *reinterpret_cast<const uint4*>(a.dout + ((size_t)b * a.t + qr) * a.c + (size_t)head * 64 + cc * 8);
    }
    const int query = q0 + n32;
    const bool q_ok = query < a.t;
    const int qrow = q_ok ? query : a.t - 1;
    const unsigned long long erow = ((unsigned long long)(b * a.heads + head) * a.t + qrow) * (unsigned long long)a.t;
    const float lse2 = a.lse2[((size_t)b * a.heads + head) * a.t + qrow], dsum = a.dsum[((size_t)b * a.heads + head) * a.t + qrow];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    s16x8 qf[4], gf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = *reinterpret_cast<const s16x8*>(qt + n32 * TA_PITCH + (16 * ks + 8 * half) * 2);
      gf[ks] = *reinterpret_cast<const s16x8*>(gt + n32 * TA_PITCH + (16 * ks + 8 * half) * 2);
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
      const char* kr = ks_ + (sub * 32 + pm) * TA_PITCH + half * 16;
      const char* vr = vs_ + (sub * 32 + pm) * TA_PITCH + half * 16;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const s16x8*>(kr + ks * 32), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const s16x8*>(vr + ks * 32), gf[ks], dp, 0, 0, 0);
      }
      attn_bwd_tile(s, dp, k0 + sub * 32 + 8 * half, lim, q_ok, lse2, dsum, a, erow, drop);
      // row = this lane's query, columns = the keys of its two runs of 8: 32 sub + 8 half + 0..7 and + 16
#pragma unroll
      for (int run = 0; run < 2; ++run) {
        const int col = (32 * sub + 16 * run + 8 * half) * 2;
        *reinterpret_cast<uint4*>(pt + n32 * TA_PITCH + col) = uint4{pack_bf16(s[8 * run + 0], s[8 * run + 1]), pack_bf16(s[8 * run + 2], s[8 * run + 3]),
                                                                      pack_bf16(s[8 * run + 4], s[8 * run + 5]), pack_bf16(s[8 * run + 6], s[8 * run + 7])};
        *reinterpret_cast<uint4*>(st + n32 * TA_PITCH + col) = uint4{pack_bf16(dp[8 * run + 0], dp[8 * run + 1]), pack_bf16(dp[8 * run + 2], dp[8 * run + 3]),
                                                                      pack_bf16(dp[8 * run + 4], dp[8 * run + 5]), pack_bf16(dp[8 * run + 6], dp[8 * run + 7])};
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- contraction over the 32 queries: two k-steps of 16
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      s16x8 ga[2], qa[2], pb[2], sb[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int off = 16 * ks * TA_PITCH + tr_off + 64 * x;
        auto tr8 = [&](const char* tile) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)tile + off));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)tile + off + 4 * TA_PITCH));
          return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        ga[x] = tr8(gt); qa[x] = tr8(qt); pb[x] = tr8(pt); sb[x] = tr8(st);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          dv[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[mt], pb[nt], dv[mt][nt], 0, 0, 0);
          dk[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[mt], sb[nt], dk[mt][nt], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  // ---- the four waves' sums: [wave][dv | dk][mt][nt][16][64 lanes] f32 over the (now idle) staging area; wave w finishes block (mt, nt) = (w >> 1, w & 1)
  __syncthreads();
  float* const red = reinterpret_cast<float*>(sm + 2 * TA_KT * TA_PITCH);      // 4 waves x 2 x 4 x 16 x 64 floats = 128 KiB: in two rounds (dv, then dk)
#pragma unroll
  for (int which = 0; which < 2; ++which) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(((size_t)wave * 4 + mt * 2 + nt) * 16 + i) * 64 + lane] = which ? dk[mt][nt][i] : dv[mt][nt][i];
    __syncthreads();
    const int mt = wave >> 1, nt = wave & 1;
    f32x16 tot;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += red[(((size_t)w * 4 + mt * 2 + nt) * 16 + i) * 64 + lane];
      tot[i] = which ? v * a.scale : v;
    }
    // accumulator register r of block (mt, nt): d = 32 mt + (r & 3) + 8 (r >> 2) + 4 half, key = k0 + 32 nt + n32
    const int key = k0 + 32 * nt + n32;
    if (key < a.t) {
      float* dst = a.dqkv + ((size_t)b * a.t + key) * rowp + (size_t)(which ? 1 : 2) * a.c + (size_t)head * 64 + 32 * mt + 4 * half;
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(dst + 8 * g) = f32x4{tot[4 * g], tot[4 * g + 1], tot[4 * g + 2], tot[4 * g + 3]};
    }
    __syncthreads();
  }
}

}  // namespace

}  // namespace ts

using namespace ts;

static int ta_check(const void* qkv, int32_t batch, int32_t t, int32_t c, int32_t heads, float p_drop) {
  if (!qkv || batch <= 0 || t <= 0 || c <= 0 || heads <= 0 || c % heads || !(p_drop >= 0.f && p_drop < 1.f)) return TS_EINVAL;
  if (c / heads != 64 || c % 8 || (reinterpret_cast<uintptr_t>(qkv) & 15) || (long long)batch * heads * t * t >= (1ll << 40)) return TS_EUNSUPPORTED;
  return TS_OK;
}

/* see include/thunder_speech_amd.h */
// words of the mask bitstring: every element + two words of slack for the 64-bit window of the last run
static long long ta_mask_words(int batch, int t, int heads) { return ((long long)batch * heads * t * t + 31) / 32 + 2; }
static void ta_draw_mask(unsigned* mask, int batch, int t, int heads, unsigned long long seed, float p, hipStream_t stream) {
  const long long n = ta_mask_words(batch, t, heads);
  hipLaunchKernelGGL(attn_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, mask, n, seed, p);
}

extern "C" int64_t ts_w2v_attention_train_fwd_workspace(int32_t batch, int32_t t, int32_t c, int32_t heads) {
  if (batch <= 0 || t <= 0 || c <= 0 || heads <= 0) return TS_EINVAL;
  return (ta_mask_words(batch, t, heads) * 4 + 15) / 16 * 16;
}

extern "C" int ts_w2v_attention_train_fwd(const void* qkv_bf16, int32_t batch, int32_t t, int32_t c, int32_t heads, const int32_t* key_len, float p_drop,
                                          uint64_t seed, float* ctx, float* lse2, void* workspace, void* stream_) {
  if (int st = ta_check(qkv_bf16, batch, t, c, heads, p_drop)) return st;
  if (!ctx || !lse2 || (reinterpret_cast<uintptr_t>(ctx) & 15) || (p_drop > 0.f && (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15)))) return TS_EINVAL;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  (void)hipGetLastError();
  if (p_drop > 0.f) ta_draw_mask(static_cast<unsigned*>(workspace), batch, t, heads, seed, p_drop, stream);
  TaArgs a{};
  a.qkv = static_cast<const unsigned short*>(qkv_bf16); a.key_len = key_len; a.ctx = ctx; a.lse2 = lse2;
  a.t = t; a.c = c; a.heads = heads;
  a.scale = 1.f / sqrtf(64.f); a.scale_log2e = 1.4426950408889634f * a.scale;
  a.p_drop = p_drop; a.keep_scale = 1.f / (1.f - p_drop); a.seed = seed;
  a.mask = p_drop > 0.f ? static_cast<const unsigned*>(workspace) : nullptr;
  hipLaunchKernelGGL(attn_fwd_train_kernel, dim3((t + TA_QW - 1) / TA_QW, heads, batch), dim3(256), 0, stream, a);
  return hip_status(hipGetLastError());
}

extern "C" int64_t ts_w2v_attention_train_bwd_workspace(int32_t batch, int32_t t, int32_t c, int32_t heads) {
  if (batch <= 0 || t <= 0 || c <= 0 || heads <= 0) return TS_EINVAL;
  return ((int64_t)batch * t * c * 2 + 15) / 16 * 16 + ((int64_t)batch * heads * t * 4 + 15) / 16 * 16 + (ta_mask_words(batch, t, heads) * 4 + 15) / 16 * 16;
}

/* see include/thunder_speech_amd.h */
extern "C" int ts_w2v_attention_train_bwd(const void* qkv_bf16, int32_t batch, int32_t t, int32_t c, int32_t heads, const int32_t* key_len, float p_drop,
                                          uint64_t seed, const float* dctx, const float* ctx, const float* lse2, const void* fwd_mask, float* dqkv, void* workspace,
                                          void* stream_) {
  if (int st = ta_check(qkv_bf16, batch, t, c, heads, p_drop)) return st;
  if (!dctx || !ctx || !lse2 || !dqkv || !workspace) return TS_EINVAL;
  if ((reinterpret_cast<uintptr_t>(dctx) & 15) || (reinterpret_cast<uintptr_t>(ctx) & 15) || (reinterpret_cast<uintptr_t>(dqkv) & 15) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15))
    return TS_EUNSUPPORTED;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  TaArgs a{};
  a.qkv = static_cast<const unsigned short*>(qkv_bf16); a.key_len = key_len; a.lse2 = const_cast<float*>(lse2); a.dqkv = dqkv;
  unsigned short* const dout16 = static_cast<unsigned short*>(workspace);
  float* const dsum = reinterpret_cast<float*>(static_cast<char*>(workspace) + ((int64_t)batch * t * c * 2 + 15) / 16 * 16);
  a.dout = dout16; a.dsum = dsum;
  a.t = t; a.c = c; a.heads = heads;
  a.scale = 1.f / sqrtf(64.f); a.scale_log2e = 1.4426950408889634f * a.scale;
  a.p_drop = p_drop; a.keep_scale = 1.f / (1.f - p_drop); a.seed = seed;
  (void)hipGetLastError();
  if (p_drop > 0.f && fwd_mask) a.mask = static_cast<const unsigned*>(fwd_mask);      // the forward's workspace, kept by the caller (4 MB per layer at 8 x 10 s)
  else if (p_drop > 0.f) {                           // or re-drawn: the mask is a pure function of the seed
    unsigned* const mask = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(dsum) + ((int64_t)batch * heads * t * 4 + 15) / 16 * 16);
    ta_draw_mask(mask, batch, t, heads, seed, p_drop, stream);
    a.mask = mask;
  }
  const long long rows = (long long)batch * t;
  hipLaunchKernelGGL(attn_rowdot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, dctx, ctx, dout16, dsum, rows, t, c, heads);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((t + TA_QW - 1) / TA_QW, heads, batch), dim3(256), 0, stream, a);
  const size_t lds = (size_t)2 * TA_KT * TA_PITCH + (size_t)4 * 4 * TA_WTILE;                  // K, V tiles + 4 waves x 4 tiles (72 KiB; one round of the final sums needs 64)
  static bool attr[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TS_EINVAL;
  if (!attr[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return TS_EUNSUPPORTED;
    attr[dev] = true;
  }
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((t + TA_KT - 1) / TA_KT, heads, batch), dim3(256), lds, stream, a);
  return hip_status(hipGetLastError());
}
