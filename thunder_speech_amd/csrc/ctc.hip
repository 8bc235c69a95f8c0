// CTC loss (forward + gradient w.r.t. the logits) for gfx950, one 256-thread workgroup per utterance.
// Replaces ctc_loss.py:36-47: permute -> log_softmax(dim=2) -> F.ctc_loss(blank, reduction="mean",
// zero_infinity=True) and its autograd backward.  Graves et al. 2006: log-space alpha/beta recursions
// over the blank-extended target (L = 2S+1 states spread over the 256 threads, rows kept in LDS), then
//   dL/dlogit[b, v, t] = (softmax[b, v, t] - sum_{s: ext[s]=v} exp(alpha + beta - lp + nll)) * g_b,
//   g_b = 1 / (B * max(S_b, 1)), zero for t >= input_len and for utterances whose loss is inf (A10).
#include "ts_common.hpp"

#include <type_traits>

namespace ts {

constexpr float NEG_INF = -__builtin_huge_valf();

__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  if (m == NEG_INF) return NEG_INF;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter, i.e. every step would wait
// for its row store to be acknowledged and for the whole emission prefetch ring
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// log(e^a + e^b + e^c) without a branch and on the bare v_exp_f32 / v_log_f32 (base 2; the library forms add denormal-range scaling
// the recursion does not need: the largest term is 1).  All three -inf: the sum is 0 and v_log_f32(0) = -inf.
__device__ __forceinline__ float lse3_fast(float a, float b, float c) {
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  const float m = fmaxf(fmaxf(a, b), c);
  const float ms = m == NEG_INF ? 0.f : m;
  const float sum = __builtin_amdgcn_exp2f((a - ms) * LOG2E) + __builtin_amdgcn_exp2f((b - ms) * LOG2E) + __builtin_amdgcn_exp2f((c - ms) * LOG2E);
  return ms + __builtin_amdgcn_logf(sum) * LN2;
}

struct CtcArgs {
  const float* logits;      // [B][V][pitch]
  const int* targets;       // [B][s_max]
  const int* input_len;
  const int* target_len;
  float* nll;               // [B]
  float* grad;              // [B][V][pitch] or null
  float* lse;               // workspace [B][T]   log-sum-exp of each frame (alpha workgroup's copy)
  float* alpha;             // workspace [B][T + 1][rowp]   (row T: scratch for the steps a partial last chunk does not run)
  float* lse2;              // workspace [B][T]   (beta workgroup's copy)
  float* beta;              // workspace [B][T + 1][rowp]
  int* feasible;            // workspace [B]
  int batch, n_classes, n_frames, pitch, s_max, lmax, blank;
  int rowp;                 // row pitch of alpha / beta: threads x states per thread (every lane owns a column, so row stores need no predicate)
};

constexpr int CTC_PF = 8;       // emission prefetch distance in time steps

// The recursions are serial in t, so the work is arranged to shorten the serial path: the alpha and the beta recursion of an
// utterance run CONCURRENTLY in two workgroups (blockIdx.y), each storing its rows, and the gradient
//   dL/dlogit[b, v, t] = (softmax - sum_{s: ext[s]=v} exp(alpha + beta - lp + nll)) * g_b
// is a third, fully parallel kernel over (utterance, frame).  Inside a step the dependent chain is kept to
//   two neighbour reads from LDS -> log-sum-exp -> LDS write -> one workgroup barrier (147 ns on this chip, tools/diag/probe_barrier.hip):
// one state per thread up to 1 024 states (round_up(2 s_max + 1, 64) threads: 2S+1 = 281 states used to cost two passes of 256 threads),
// the thread keeps its own previous value in a register, the per-frame log-sum-exp comes out of LDS, and the emission
// lg[ext[s]][t] -- its address depends on (s, t) only -- arrives in chunks of CTC_PF steps requested a whole chunk ahead, the rows leave
// in chunks too: no vector-memory operation and no vmcnt wait inside a chunk.  (__syncthreads() would drain vmcnt at every step;
// a load under a condition becomes a phi the compiler waits on at once: both measured, 0.78 us per step before, 0.5 with them.)
// LSE_LDS: the per-frame log-sum-exp row is kept in LDS (n_frames floats: up to ~38 000 frames); false: clips longer than that read it back from the
// global copy -- prefetched with the emissions, a chunk ahead, so the step loop still contains no vector-memory wait
template <int SPT, bool LSE_LDS = true>
__global__ __launch_bounds__(1024) void ctc_kernel(const CtcArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int L_MAX = a.lmax;
  float* const row0 = reinterpret_cast<float*>(smem);          // [lmax + 4]: two -inf guard cells on either side
  float* const row1 = row0 + L_MAX + 4;
  int* const ext = reinterpret_cast<int*>(row1 + L_MAX + 4);   // [lmax]
  float* const lse_s = reinterpret_cast<float*>(ext + L_MAX);  // [n_frames]
  const int b = blockIdx.x, lane = threadIdx.x;
  const bool backward = blockIdx.y != 0;
  const int V = a.n_classes;
  int T = a.input_len[b];
  T = T < 0 ? 0 : (T > a.n_frames ? a.n_frames : T);
  int S = a.target_len[b];
  S = S < 0 ? 0 : (S > a.s_max ? a.s_max : S);
  const int L = 2 * S + 1;
  const float* lg = a.logits + (size_t)b * V * a.pitch;
  float* const lse = (backward ? a.lse2 : a.lse) + (size_t)b * a.n_frames;        // each direction keeps its own copy
  float* const rows = (backward ? a.beta : a.alpha) + (size_t)b * (a.n_frames + 1) * a.rowp;

  for (int s = lane; s < L; s += blockDim.x) ext[s] = (s & 1) ? a.targets[(size_t)b * a.s_max + (s >> 1)] : a.blank;
  for (int s = lane; s < L_MAX + 4; s += blockDim.x) { row0[s] = NEG_INF; row1[s] = NEG_INF; }
  // log-sum-exp per frame (coalesced over t), online over the classes in batches of 8 INDEPENDENT loads: a plain loop over v is a
  // chain of V dependent round trips to memory per pass (2 x 29 x ~0.7 us before the recursion has even started)
  for (int t = lane; t < T; t += blockDim.x) {
    float m = NEG_INF, sum = 0.f;
    for (int v0 = 0; v0 < V; v0 += 8) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = lg[(size_t)min(v0 + u, V - 1) * a.pitch + t];
      float bm = x[0];
#pragma unroll
      for (int u = 1; u < 8; ++u) bm = fmaxf(bm, x[u]);
      const float mn = fmaxf(m, bm);
      float part = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) part += (v0 + u < V) ? expf(x[u] - mn) : 0.f;
      sum = sum * expf(m - mn) + part;             // m = -inf on the first batch: exp(-inf) = 0
      m = mn;
    }
    const float l = m + logf(sum);
    lse[t] = l;
    if constexpr (LSE_LDS) lse_s[t] = l;
  }
  __syncthreads();
  float* prev = row0 + 2;
  float* cur = row1 + 2;
  const int d = backward ? 1 : -1;                 // neighbour direction: alpha looks at s-1, s-2, beta at s+1, s+2

  constexpr int NS = SPT;
  const int NT = blockDim.x;                       // round_up(lmax / SPT, 64) threads: state s = lane + i * NT
  bool act[NS], skip[NS], init[NS];
  const float* lrow[NS];
  float own[NS], pf[2][CTC_PF][NS], out[CTC_PF][NS], lq[2][LSE_LDS ? 1 : CTC_PF];
  // step k of this direction looks at frame fr(k); emissions are requested a whole chunk of CTC_PF steps ahead, for lanes without a
  // state from the blank's row and with clamped frame indices (all loads unconditional, no vector-memory operation inside a chunk:
  // a load under a condition turns the ring slot into a phi and the compiler then waits for the load it has just issued)
  auto fr = [&](int k) { k = min(k, T - 1); return backward ? T - 1 - k : k; };
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int s = lane + i * NT;
    act[i] = s < L;
    const int e = act[i] ? ext[s] : a.blank;
    const int s2 = s + 2 * d;
    // the skip transition: from a different non-blank label two states away (alpha: into s, beta: out of s)
    skip[i] = act[i] && s2 >= 0 && s2 < L && (backward ? (ext[s2] != a.blank && ext[s2] != e) : (e != a.blank && e != ext[s2]));
    init[i] = act[i] && (backward ? s >= L - 2 : s < 2);
    lrow[i] = lg + (size_t)e * a.pitch;
    own[i] = NEG_INF;
  }
  auto fetch = [&](float (&buf)[CTC_PF][NS], float (&lb)[LSE_LDS ? 1 : CTC_PF], int k0) {
#pragma unroll
    for (int j = 0; j < CTC_PF; ++j)
#pragma unroll
      for (int i = 0; i < NS; ++i) buf[j][i] = lrow[i][fr(k0 + j)];
    if constexpr (!LSE_LDS) {
#pragma unroll
      for (int j = 0; j < CTC_PF; ++j) lb[j] = lse[fr(k0 + j)];
    }
  };
  auto chunk = [&](float (&buf)[CTC_PF][NS], float (&lb)[LSE_LDS ? 1 : CTC_PF], int k0) {
#pragma unroll
    for (int j = 0; j < CTC_PF; ++j) {
      const int k = k0 + j;
      if (k < T) {                                 // workgroup-uniform
        float n1[NS], n2[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {             // all LDS reads of the step first: the slots' chains then interleave
          const int s = act[i] ? lane + i * NT : 0;
          n1[i] = prev[s + d];
          n2[i] = prev[s + 2 * d];
        }
        float ls;
        if constexpr (LSE_LDS) ls = lse_s[fr(k)]; else ls = lb[j];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const float lpv = buf[j][i] - ls;
          float v = lse3_fast(own[i], n1[i], skip[i] ? n2[i] : NEG_INF) + lpv;
          if (k == 0) v = init[i] ? lpv : NEG_INF;
          own[i] = v;
          out[j][i] = v;
        }
#pragma unroll
        for (int i = 0; i < NS; ++i)
          if (act[i]) cur[lane + i * NT] = own[i];
        lds_barrier();
        float* tmp = prev; prev = cur; cur = tmp;
      }
    }
    // the chunk's rows go out together, then the emissions of the chunk after next are requested into the buffer just used
#pragma unroll
    for (int j = 0; j < CTC_PF; ++j)
#pragma unroll
      for (int i = 0; i < NS; ++i)
        rows[(size_t)(k0 + j < T ? fr(k0 + j) : a.n_frames) * a.rowp + lane + i * NT] = out[j][i];     // unpredicated: see rowp
    fetch(buf, lb, k0 + 2 * CTC_PF);
  };
  if (T > 0) {
    fetch(pf[0], lq[0], 0);
    fetch(pf[1], lq[1], CTC_PF);
    for (int k0 = 0; k0 < T; k0 += 2 * CTC_PF) {
      chunk(pf[0], lq[0], k0);
      chunk(pf[1], lq[1], k0 + CTC_PF);                   // (its steps are guarded by k < T; its loads and stores always run: exact vmcnt counts)
    }
  }
  if (backward) return;
  bool feasible;
  float nll = 0.f;
  if (T == 0) {
    feasible = (S == 0);
  } else {
    const float l1 = prev[L - 1], l2 = L > 1 ? prev[L - 2] : NEG_INF;
    const float ll = lse3(l1, l2, NEG_INF);
    feasible = ll > NEG_INF;
    nll = -ll;
  }
  if (lane == 0) {
    a.nll[b] = feasible ? nll : 0.f;               // zero_infinity=True
    a.feasible[b] = feasible ? 1 : 0;
  }
}

// gradient: one wavefront per (utterance, frame), 4 frames per workgroup
__global__ __launch_bounds__(256) void ctc_grad_kernel(const CtcArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int L_MAX = a.lmax, V = a.n_classes;
  int* const ext = reinterpret_cast<int*>(smem);                         // [lmax]
  float* const occ_all = reinterpret_cast<float*>(ext + L_MAX);          // [4][V]
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + wave;
  int T = a.input_len[b];
  T = T < 0 ? 0 : (T > a.n_frames ? a.n_frames : T);
  int S = a.target_len[b];
  S = S < 0 ? 0 : (S > a.s_max ? a.s_max : S);
  const int L = 2 * S + 1;
  for (int s = threadIdx.x; s < L; s += 256) ext[s] = (s & 1) ? a.targets[(size_t)b * a.s_max + (s >> 1)] : a.blank;
  float* const occ = occ_all + wave * V;
  for (int v = lane; v < V; v += 64) occ[v] = 0.f;
  __syncthreads();
  if (t >= a.n_frames) return;
  float* gb = a.grad + (size_t)b * V * a.pitch;
  const float* lg = a.logits + (size_t)b * V * a.pitch;
  if (t >= T || !a.feasible[b]) {                                        // frames >= T, and everything when the loss is inf
    for (int v = lane; v < V; v += 64) gb[(size_t)v * a.pitch + t] = 0.f;
    return;
  }
  const float scale = 1.f / ((float)a.batch * (float)(S > 0 ? S : 1));
  const float l = a.lse[(size_t)b * a.n_frames + t];
  const float nll = a.nll[b];
  const float* al = a.alpha + ((size_t)b * (a.n_frames + 1) + t) * a.rowp;
  const float* be = a.beta + ((size_t)b * (a.n_frames + 1) + t) * a.rowp;
  float blank_w = 0.f;
  for (int s = lane; s < L; s += 64) {
    const int e = ext[s];
    const float w = expf(al[s] + be[s] - (lg[(size_t)e * a.pitch + t] - l) + nll);
    if (s & 1) atomicAdd(&occ[e], w);                                    // a label: few states share a class
    else blank_w += w;                                                   // the blank: every other state -> reduce in registers
  }
  for (int o = 32; o > 0; o >>= 1) blank_w += __shfl_xor(blank_w, o);
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) atomicAdd(&occ[a.blank], blank_w);
  __builtin_amdgcn_s_waitcnt(0xc07f);                                    // lgkmcnt(0): this wave's LDS atomics have landed
  for (int v = lane; v < V; v += 64) gb[(size_t)v * a.pitch + t] = (expf(lg[(size_t)v * a.pitch + t] - l) - occ[v]) * scale;
}

// Targets / lengths as F.ctc_loss accepts them -> what ctc_kernel indexes with, in ONE launch (the same rules spelled in torch ops
// were 13 launches of a fine-tuning step): padded labels int32 with positions >= length and wild ids rewritten to 0; an utterance
// holding a wild id (outside [0, V)) at a position < its length is made infeasible -- input length 0, target length >= 1 -- so its
// loss is the infinity zero_infinity turns into 0 and its gradient is 0; lengths are truncated toward zero like `.long()` (A5).
// kind: 0 int32, 1 int64, 2 float32, 3 float64.
__device__ __forceinline__ long long ctc_read_len(const void* p, int kind, int i) {
  switch (kind) {
    case 0: return static_cast<const int*>(p)[i];
    case 1: return static_cast<const long long*>(p)[i];
    case 2: return (long long)static_cast<const float*>(p)[i];
    default: return (long long)static_cast<const double*>(p)[i];
  }
}

__global__ __launch_bounds__(256) void ctc_prepare_kernel(const void* targets, int tg_kind, long long tg_stride, const void* target_len, int tl_kind,
                                                          const void* input_len, int il_kind, int s_in, int s_max, int n_classes,
                                                          int* tg_out, int* tl_out, int* il_out, int* bad_total) {
  const int b = blockIdx.x;
  long long tl = ctc_read_len(target_len, tl_kind, b);
  const long long tlc = tl < 0 ? 0 : (tl > s_in ? s_in : tl);
  int bad = 0;
  for (int j = threadIdx.x; j < s_max; j += 256) {
    long long id = 0;
    if (j < tlc) id = tg_kind == 1 ? static_cast<const long long*>(targets)[(size_t)b * tg_stride + j] : static_cast<const int*>(targets)[(size_t)b * tg_stride + j];
    const bool wild = id < 0 || id >= n_classes;
    bad |= wild ? 1 : 0;
    tg_out[(size_t)b * s_max + j] = wild ? 0 : (int)id;
  }
  bad = __syncthreads_or(bad);
  if (threadIdx.x == 0) {
    long long il = ctc_read_len(input_len, il_kind, b);
    il = il < -2147483647LL ? -2147483647LL : (il > 2147483647LL ? 2147483647LL : il);
    tl = tl < -2147483647LL ? -2147483647LL : (tl > 2147483647LL ? 2147483647LL : tl);
    il_out[b] = bad ? 0 : (int)il;
    tl_out[b] = (bad && tl == 0) ? 1 : (int)tl;
    if (bad && bad_total) atomicAdd(bad_total, 1);                            // running count of dropped utterances (the caller's to check)
  }
}

__global__ void ctc_mean_kernel(const float* nll, const int* target_len, int batch, int s_max, float* loss) {
  // reduction="mean": mean over the batch of nll / clamp(target_len, 1)
  float s = 0.f;
  for (int b = threadIdx.x; b < batch; b += 64) {
    int S = target_len[b];
    S = S < 1 ? 1 : (S > s_max ? s_max : S);
    s += nll[b] / (float)S;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if (threadIdx.x == 0) loss[0] = s / (float)batch;
}

}  // namespace ts

static int ctc_states_per_thread(int lmax) { return lmax <= 1024 ? 1 : (lmax <= 2048 ? 2 : 4); }
static int ctc_threads(int lmax) { const int spt = ctc_states_per_thread(lmax); return ((lmax + spt - 1) / spt + 63) / 64 * 64; }

extern "C" int64_t ts_ctc_workspace_bytes(int32_t batch, int32_t n_classes, int32_t n_frames, int32_t s_max) {
  (void)n_classes;
  if (batch <= 0 || n_frames <= 0 || s_max < 0) return TS_EINVAL;
  const int lmax = 2 * s_max + 1;
  const int64_t rowp = (int64_t)ctc_states_per_thread(lmax) * ctc_threads(lmax);
  return 2 * ((int64_t)batch * n_frames * sizeof(float) + (int64_t)batch * (n_frames + 1) * rowp * sizeof(float)) + (int64_t)batch * sizeof(int);
}

extern "C" int ts_ctc_loss(const float* logits, int32_t batch, int32_t n_classes, int32_t n_frames, int32_t pitch,
                           const int32_t* targets, int32_t s_max, const int32_t* input_len, const int32_t* target_len,
                           int32_t blank, float* nll, float* loss, float* grad, void* workspace, void* stream_) {
  using namespace ts;
  if (!logits || !targets || !input_len || !target_len || !nll || !loss || !workspace) return TS_EINVAL;
  if (batch <= 0 || n_classes <= 0 || n_frames <= 0 || pitch < n_frames || s_max < 0) return TS_EINVAL;
  if (blank < 0 || blank >= n_classes) return TS_EINVAL;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  CtcArgs a{};
  a.logits = logits; a.targets = targets; a.input_len = input_len; a.target_len = target_len;
  a.nll = nll; a.grad = grad;
  a.lse = static_cast<float*>(workspace);
  a.batch = batch; a.n_classes = n_classes; a.n_frames = n_frames; a.pitch = pitch; a.s_max = s_max;
  a.lmax = 2 * s_max + 1; a.blank = blank;
  const int spt = ctc_states_per_thread(a.lmax);
  a.rowp = spt * ctc_threads(a.lmax);
  a.alpha = a.lse + (size_t)batch * n_frames;
  a.lse2 = a.alpha + (size_t)batch * (n_frames + 1) * a.rowp;
  a.beta = a.lse2 + (size_t)batch * n_frames;
  a.feasible = reinterpret_cast<int*>(a.beta + (size_t)batch * (n_frames + 1) * a.rowp);
  size_t lds = ((size_t)2 * (a.lmax + 4) + a.lmax + (size_t)n_frames) * sizeof(float);
  const size_t lds_g = ((size_t)a.lmax + 4 * (size_t)n_classes) * sizeof(float);
  // the per-frame log-sum-exp row lives in LDS next to the two state rows: 160 KiB hold about 38 000 frames (a 12-minute clip after the stem);
  // longer clips take the instantiation that reads the row back from its global copy
  const bool lse_lds = lds <= 160 * 1024;
  if (!lse_lds) lds = ((size_t)2 * (a.lmax + 4) + a.lmax) * sizeof(float);
  if (lds > 160 * 1024 || lds_g > 64 * 1024 || a.lmax > 4096) return TS_EUNSUPPORTED;            // more than 2 047 labels in a transcript
  (void)hipGetLastError();
  const dim3 grid(batch, grad ? 2 : 1);                                                          // alpha || beta
  const dim3 block((unsigned)ctc_threads(a.lmax));
  static bool big_all[64][3] = {};                                                               // more than the default 64 KiB allowed: per (device, instantiation)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TS_EINVAL;
  bool (&big)[3] = big_all[dev];
  const int which = spt == 1 ? 0 : (spt == 2 ? 1 : 2);
  if (!lse_lds) {
    if (which == 0) hipLaunchKernelGGL((ctc_kernel<1, false>), grid, block, lds, stream, a);
    else if (which == 1) hipLaunchKernelGGL((ctc_kernel<2, false>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((ctc_kernel<4, false>), grid, block, lds, stream, a);
  } else {
  if (lds > 64 * 1024 && !big[which]) {
    const void* fn = which == 0 ? reinterpret_cast<const void*>(ctc_kernel<1>)
                                : (which == 1 ? reinterpret_cast<const void*>(ctc_kernel<2>) : reinterpret_cast<const void*>(ctc_kernel<4>));
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return TS_EUNSUPPORTED;
    big[which] = true;
  }
  if (which == 0) hipLaunchKernelGGL(ctc_kernel<1>, grid, block, lds, stream, a);
  else if (which == 1) hipLaunchKernelGGL(ctc_kernel<2>, grid, block, lds, stream, a);
  else hipLaunchKernelGGL(ctc_kernel<4>, grid, block, lds, stream, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (grad) hipLaunchKernelGGL(ctc_grad_kernel, dim3((n_frames + 3) / 4, batch), dim3(256), lds_g, stream, a);
  (void)hipGetLastError();
  hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(64), 0, stream, nll, target_len, batch, s_max, loss);
  return hip_status(hipGetLastError());
}

extern "C" int ts_ctc_prepare(const void* targets, int32_t targets_kind, int64_t targets_stride, int32_t s_in, const void* target_len,
                              int32_t target_len_kind, const void* input_len, int32_t input_len_kind, int32_t batch, int32_t s_max,
                              int32_t n_classes, int32_t* targets_out, int32_t* target_len_out, int32_t* input_len_out, int32_t* bad_rows_total,
                              void* stream) {
  if (!targets || !target_len || !input_len || !targets_out || !target_len_out || !input_len_out) return TS_EINVAL;
  if (batch <= 0 || s_max <= 0 || s_in < 0 || s_in > s_max || n_classes <= 0 || targets_stride < s_in) return TS_EINVAL;
  if (targets_kind < 0 || targets_kind > 1 || target_len_kind < 0 || target_len_kind > 3 || input_len_kind < 0 || input_len_kind > 3) return TS_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::ctc_prepare_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, targets, targets_kind, (long long)targets_stride, target_len,
                     target_len_kind, input_len, input_len_kind, s_in, s_max, n_classes, targets_out, target_len_out, input_len_out, bad_rows_total);
  return ts::hip_status(hipGetLastError());
}
