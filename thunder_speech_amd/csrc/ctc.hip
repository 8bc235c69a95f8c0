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
  float* lse;               // workspace [B][T]   log-sum-exp of each frame
  float* alpha;             // workspace [B][T][lmax]
  int batch, n_classes, n_frames, pitch, s_max, lmax, blank;
};

// The recursions are serial in t, so a step must stay short.  Measured: staging the emissions / alpha rows in LDS chunks does NOT
// help (the gathers overlap with the log-sum-exp arithmetic of the step); what a backward step spent its time on was (a) the LDS
// atomics of the occupancy sums -- every other state is the blank, ~S atomics on ONE address -- now a wavefront reduction, and
// (b) the dense softmax term of the gradient (V exps and scattered stores per step), now one coalesced sweep after the loop.
constexpr int CTC_NT = 256;     // threads per utterance: one state per thread up to S = 127, so a step is one pass, not L / 64

__global__ __launch_bounds__(CTC_NT) void ctc_kernel(const CtcArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int L_MAX = a.lmax;
  float* const row0 = reinterpret_cast<float*>(smem);          // [lmax + 2] (two leading -inf guards)
  float* const row1 = row0 + L_MAX + 2;
  int* const ext = reinterpret_cast<int*>(row1 + L_MAX + 2);   // [lmax]
  float* const occ = reinterpret_cast<float*>(ext + L_MAX);    // [n_classes]
  const int b = blockIdx.x, lane = threadIdx.x;
  const int V = a.n_classes;
  int T = a.input_len[b];
  T = T < 0 ? 0 : (T > a.n_frames ? a.n_frames : T);
  int S = a.target_len[b];
  S = S < 0 ? 0 : (S > a.s_max ? a.s_max : S);
  const int L = 2 * S + 1;
  const float* lg = a.logits + (size_t)b * V * a.pitch;
  float* const lse = a.lse + (size_t)b * a.n_frames;
  float* const alpha = a.alpha + (size_t)b * a.n_frames * L_MAX;

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

  // ---- alpha ------------------------------------------------------------------------------------------
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
        alpha[(size_t)t * L_MAX + s] = v;
      }
      __syncthreads();
      float* tmp = prev; prev = cur; cur = tmp;
    }
    const float l1 = prev[L - 1], l2 = L > 1 ? prev[L - 2] : NEG_INF;
    const float ll = lse3(l1, l2, NEG_INF);
    feasible = ll > NEG_INF;
    nll = -ll;
  }
  if (!feasible) nll = 0.f;                          // zero_infinity=True
  if (lane == 0) a.nll[b] = nll;
  if (!a.grad) return;

  // ---- beta + gradient ---------------------------------------------------------------------------------
  float* gb = a.grad + (size_t)b * V * a.pitch;
  const float scale = 1.f / ((float)a.batch * (float)(S > 0 ? S : 1));
  // frames >= T (and everything when infeasible): zero gradient
  for (int v = 0; v < V; ++v)
    for (int t = (feasible ? T : 0) + lane; t < a.n_frames; t += CTC_NT) gb[(size_t)v * a.pitch + t] = 0.f;
  if (!feasible || T == 0) return;
  for (int v = lane; v < V; v += CTC_NT) occ[v] = 0.f;
  __syncthreads();
  // beta rows live in prev/cur with two trailing guards: use index s+1, s+2 < L checks instead
  for (int t = T - 1; t >= 0; --t) {
    float blank_w = 0.f;                                      // occupancy of the blank states handled by this lane
    for (int s = lane; s < L; s += CTC_NT) {
      const float e_lp = lp(t, s);
      float v;
      if (t == T - 1) {
        v = (s >= L - 2) ? e_lp : NEG_INF;
      } else {
        const int e = ext[s];
        const float b0 = prev[s];
        const float b1 = s + 1 < L ? prev[s + 1] : NEG_INF;
        const float b2 = (s + 2 < L && ext[s + 2] != a.blank && ext[s + 2] != e) ? prev[s + 2] : NEG_INF;
        v = lse3(b0, b1, b2) + e_lp;
      }
      cur[s] = v;
      // occupancy exp(alpha + beta - lp + nll): v already holds beta[t][s]
      const float w = expf(alpha[(size_t)t * L_MAX + s] + v - e_lp + nll);
      if (s & 1) atomicAdd(&occ[ext[s]], w);                  // a label: few states share a class
      else blank_w += w;                                      // the blank: every other state -> reduce in registers
    }
    for (int o = 32; o > 0; o >>= 1) blank_w += __shfl_xor(blank_w, o);
    if ((lane & 63) == 0) atomicAdd(&occ[a.blank], blank_w);       // one atomic per wavefront
    __syncthreads();
    for (int v = lane; v < V; v += CTC_NT) {
      gb[(size_t)v * a.pitch + t] = -occ[v] * scale;          // the softmax term follows below
      occ[v] = 0.f;
    }
    __syncthreads();
    float* tmp = prev; prev = cur; cur = tmp;
  }
  __syncthreads();
  // dense part of the gradient, coalesced over t: + softmax * scale
  for (int t = lane; t < T; t += CTC_NT) {
    const float l = lse[t];
    for (int v = 0; v < V; ++v) gb[(size_t)v * a.pitch + t] += expf(lg[(size_t)v * a.pitch + t] - l) * scale;
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

extern "C" int64_t ts_ctc_workspace_bytes(int32_t batch, int32_t n_classes, int32_t n_frames, int32_t s_max) {
  (void)n_classes;
  if (batch <= 0 || n_frames <= 0 || s_max < 0) return TS_EINVAL;
  const int64_t lmax = 2 * (int64_t)s_max + 1;
  return (int64_t)batch * n_frames * sizeof(float) + (int64_t)batch * n_frames * lmax * sizeof(float);
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
  a.batch = batch; a.n_classes = n_classes; a.n_frames = n_frames; a.pitch = pitch; a.s_max = s_max;
  a.lmax = 2 * s_max + 1; a.blank = blank;
  const size_t lds = ((size_t)2 * (a.lmax + 2) + a.lmax + n_classes) * sizeof(float);
  if (lds > 64 * 1024) return TS_EUNSUPPORTED;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ctc_kernel, dim3(batch), dim3(CTC_NT), lds, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(64), 0, stream, nll, target_len, batch, s_max, loss);
  return hip_status(hipGetLastError());
}
