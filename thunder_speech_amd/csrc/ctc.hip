// CTC loss (forward + gradient w.r.t. the logits) for gfx950, one 256-thread workgroup per utterance.
// Replaces ctc_loss.py:36-47: permute -> log_softmax(dim=2) -> F.ctc_loss(blank, reduction="mean",
// zero_infinity=True) and its autograd backward.  Graves et al. 2006: log-space alpha/beta recursions
// over the blank-extended target (L = 2S+1 states spread over the 256 threads, rows kept in LDS), then
//   dL/dlogit[b, v, t] = (softmax[b, v, t] - sum_{s: ext[s]=v} exp(alpha + beta - lp + nll)) * g_b,
//   g_b = 1 / (B * max(S_b, 1)), zero for t >= input_len and for utterances whose loss is inf (A10).
#include "ts_common.hpp"

namespace ts {

constexpr float NEG_INF = -__builtin_huge_valf();

__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  if (m == NEG_INF) return NEG_INF;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

struct CtcArgs {
  const float* logits;      // [B][V][pitch]
  const int* targets;       // [B][s_max]
  const int* input_len;
  const int* target_len;
  float* nll;               // [B]
  float* grad;              // [B][V][pitch] or null
  float* lse;               // workspace [B][T]   log-sum-exp of each frame (alpha workgroup's copy)
  float* alpha;             // workspace [B][T][lmax]
  float* lse2;              // workspace [B][T]   (beta workgroup's copy)
  float* beta;              // workspace [B][T][lmax]
  int* feasible;            // workspace [B]
  int batch, n_classes, n_frames, pitch, s_max, lmax, blank;
};

constexpr int CTC_NT = 256;     // threads per utterance and direction: one state per thread up to S = 127

// The recursions are serial in t (one step costs about 1.2 us whatever is prefetched or staged: it is the barrier + LDS
// round trip + log-sum-exp chain), so the work is arranged to shorten the serial path instead: the alpha and the beta
// recursion of an utterance run CONCURRENTLY in two workgroups (blockIdx.y), each storing its rows, and the gradient
//   dL/dlogit[b, v, t] = (softmax - sum_{s: ext[s]=v} exp(alpha + beta - lp + nll)) * g_b
// is a third, fully parallel kernel over (utterance, frame).
__global__ __launch_bounds__(CTC_NT) void ctc_kernel(const CtcArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int L_MAX = a.lmax;
  float* const row0 = reinterpret_cast<float*>(smem);          // [lmax + 2] (two leading -inf guards)
  float* const row1 = row0 + L_MAX + 2;
  int* const ext = reinterpret_cast<int*>(row1 + L_MAX + 2);   // [lmax]
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
  float* const rows = (backward ? a.beta : a.alpha) + (size_t)b * a.n_frames * L_MAX;

  for (int s = lane; s < L; s += CTC_NT) ext[s] = (s & 1) ? a.targets[(size_t)b * a.s_max + (s >> 1)] : a.blank;
  // log-sum-exp per frame (coalesced over t)
  for (int t = lane; t < T; t += CTC_NT) {
    float m = NEG_INF;
    for (int v = 0; v < V; ++v) m = fmaxf(m, lg[(size_t)v * a.pitch + t]);
    float sum = 0.f;
    for (int v = 0; v < V; ++v) sum += expf(lg[(size_t)v * a.pitch + t] - m);
    lse[t] = m + logf(sum);
  }
  if (lane < 2) { row0[lane] = NEG_INF; row1[lane] = NEG_INF; }
  __syncthreads();
  float* prev = row0 + 2;
  float* cur = row1 + 2;
  auto lp = [&](int t, int s) { return lg[(size_t)ext[s] * a.pitch + t] - lse[t]; };

  if (!backward) {
    // ---- alpha ----------------------------------------------------------------------------------------
    float nll = 0.f;
    bool feasible = true;
    if (T == 0) {
      feasible = (S == 0);
    } else {
      for (int t = 0; t < T; ++t) {
        for (int s = lane; s < L; s += CTC_NT) {
          float v;
          if (t == 0) {
            v = s < 2 ? lp(0, s) : NEG_INF;
          } else {
            const int e = ext[s];
            const float a0 = prev[s], a1 = prev[s - 1];
            const float a2 = (s >= 2 && e != a.blank && e != ext[s - 2]) ? prev[s - 2] : NEG_INF;
            v = lse3(a0, a1, a2) + lp(t, s);
          }
          cur[s] = v;
          rows[(size_t)t * L_MAX + s] = v;
        }
        __syncthreads();
        float* tmp = prev; prev = cur; cur = tmp;
      }
      const float l1 = prev[L - 1], l2 = L > 1 ? prev[L - 2] : NEG_INF;
      const float ll = lse3(l1, l2, NEG_INF);
      feasible = ll > NEG_INF;
      nll = -ll;
    }
    if (lane == 0) {
      a.nll[b] = feasible ? nll : 0.f;               // zero_infinity=True
      a.feasible[b] = feasible ? 1 : 0;
    }
    return;
  }
  // ---- beta (rows only: trailing guards are emulated with s + 1, s + 2 < L checks) -------------------------
  for (int t = T - 1; t >= 0; --t) {
    for (int s = lane; s < L; s += CTC_NT) {
      float v;
      if (t == T - 1) {
        v = (s >= L - 2) ? lp(t, s) : NEG_INF;
      } else {
        const int e = ext[s];
        const float b0 = prev[s];
        const float b1 = s + 1 < L ? prev[s + 1] : NEG_INF;
        const float b2 = (s + 2 < L && ext[s + 2] != a.blank && ext[s + 2] != e) ? prev[s + 2] : NEG_INF;
        v = lse3(b0, b1, b2) + lp(t, s);
      }
      cur[s] = v;
      rows[(size_t)t * L_MAX + s] = v;
    }
    __syncthreads();
    float* tmp = prev; prev = cur; cur = tmp;
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
  const float* al = a.alpha + ((size_t)b * a.n_frames + t) * L_MAX;
  const float* be = a.beta + ((size_t)b * a.n_frames + t) * L_MAX;
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

extern "C" int64_t ts_ctc_workspace_bytes(int32_t batch, int32_t n_classes, int32_t n_frames, int32_t s_max) {
  (void)n_classes;
  if (batch <= 0 || n_frames <= 0 || s_max < 0) return TS_EINVAL;
  const int64_t lmax = 2 * (int64_t)s_max + 1;
  return 2 * ((int64_t)batch * n_frames * sizeof(float) + (int64_t)batch * n_frames * lmax * sizeof(float)) + (int64_t)batch * sizeof(int);
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
  a.alpha = a.lse + (size_t)batch * n_frames;
  a.lse2 = a.alpha + (size_t)batch * n_frames * (2 * (size_t)s_max + 1);
  a.beta = a.lse2 + (size_t)batch * n_frames;
  a.feasible = reinterpret_cast<int*>(a.beta + (size_t)batch * n_frames * (2 * (size_t)s_max + 1));
  a.batch = batch; a.n_classes = n_classes; a.n_frames = n_frames; a.pitch = pitch; a.s_max = s_max;
  a.lmax = 2 * s_max + 1; a.blank = blank;
  const size_t lds = ((size_t)2 * (a.lmax + 2) + a.lmax) * sizeof(float);
  const size_t lds_g = ((size_t)a.lmax + 4 * (size_t)n_classes) * sizeof(float);
  if (lds > 64 * 1024 || lds_g > 64 * 1024) return TS_EUNSUPPORTED;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ctc_kernel, dim3(batch, grad ? 2 : 1), dim3(CTC_NT), lds, stream, a);      // alpha || beta
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (grad) hipLaunchKernelGGL(ctc_grad_kernel, dim3((n_frames + 3) / 4, batch), dim3(256), lds_g, stream, a);
  (void)hipGetLastError();
  hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(64), 0, stream, nll, target_len, batch, s_max, loss);
  return hip_status(hipGetLastError());
}
