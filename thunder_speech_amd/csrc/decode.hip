// Greedy CTC decode for gfx950: per-frame argmax over the classes, then run-collapse of each row.
// Replaces module.py:100 (pred.argmax(1)) and the per-row torch.unique_consecutive Python loop of
// text_processing/transform.py:107-110.  All T' frames are decoded (out_lengths are ignored, quirk A9);
// blank removal and the id -> token mapping stay on the host (string work).
#include "ts_common.hpp"

namespace ts {

// one workgroup per clip; frames are processed in chunks of up to 1024 (one chunk covers a 20 s clip at the encoder's frame rate, so the
// class loop's loads are all in flight at once) with a running output offset.  A chunk shorter than the workgroup (Citrinet: 251 frames,
// 1 024 classes) splits the CLASSES over the idle threads instead: thread = (frame, class slice), partial maxima combined through LDS in
// slice order, so the lowest index still wins ties (61 -> 20 us at 32 x 251 x 1 024).
constexpr int GTH = 1024, GW = GTH / 64;
__global__ __launch_bounds__(GTH) void greedy_kernel(const float* __restrict__ logits, int n_classes, int n_frames, int pitch,
                                                     int* __restrict__ ids, int* __restrict__ collapsed,
                                                     int* __restrict__ counts) {
  __shared__ int wave_sum[GW];
  __shared__ int carry_s;
  __shared__ int last_id_s;
  __shared__ float part_v[GTH];
  __shared__ int part_i[GTH];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* base = logits + (size_t)b * n_classes * pitch;
  if (tid == 0) { carry_s = 0; last_id_s = -1; }
  __syncthreads();
  const int cf = n_frames >= GTH ? GTH : (n_frames + 63) / 64 * 64;            // frames per chunk
  const int n_sl = GTH / cf;                                                    // class slices
  const int ft = tid % cf, sl = tid / cf;
  const int per = (n_classes + n_sl - 1) / n_sl;
  const int v_lo = sl * per, v_hi = min(n_classes, v_lo + per);
  for (int t0 = 0; t0 < n_frames; t0 += cf) {
    const int t = t0 + ft;
    int best = 0;
    if (n_sl == 1) {
      if (t < n_frames) {
        float bv = base[t];
        int v = 1;
        for (; v + 8 <= n_classes; v += 8) {           // coalesced over t; lowest index wins ties; 8 loads in flight
          float x[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = base[(size_t)(v + j) * pitch + t];
#pragma unroll
          for (int j = 0; j < 8; ++j) if (x[j] > bv) { bv = x[j]; best = v + j; }
        }
        for (; v < n_classes; ++v) {
          const float x = base[(size_t)v * pitch + t];
          if (x > bv) { bv = x; best = v; }
        }
      }
    } else {
      float bv = -__builtin_huge_valf();
      int bi = 0x7fffffff;
      if (t < n_frames && sl < n_sl && v_lo < v_hi) {
        bv = base[(size_t)v_lo * pitch + t];
        bi = v_lo;
        int v = v_lo + 1;
        for (; v + 8 <= v_hi; v += 8) {
          float x[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = base[(size_t)(v + j) * pitch + t];
#pragma unroll
          for (int j = 0; j < 8; ++j) if (x[j] > bv) { bv = x[j]; bi = v + j; }
        }
        for (; v < v_hi; ++v) {
          const float x = base[(size_t)v * pitch + t];
          if (x > bv) { bv = x; bi = v; }
        }
      }
      part_v[tid] = bv;
      part_i[tid] = bi;
      __syncthreads();
      if (sl == 0 && t < n_frames) {
        for (int g = 1; g < n_sl; ++g) {
          const float x = part_v[g * cf + ft];
          if (x > bv) { bv = x; bi = part_i[g * cf + ft]; }                     // strict: an earlier slice keeps a tie, as torch.argmax does
        }
        best = bi;
      }
      __syncthreads();
    }
    const bool owner = sl == 0 && t < n_frames;                                 // one thread per frame from here on
    if (owner) ids[(size_t)b * n_frames + t] = best;
    // previous frame's id: neighbour lane, or the last id of the previous chunk
    int prev = __shfl_up(best, 1);
    __shared__ int edge[GW];
    if (lane == 63) edge[wave] = best;
    __syncthreads();
    if (lane == 0) prev = wave == 0 ? last_id_s : edge[wave - 1];
    const int keep = owner && (best != prev);
    // exclusive scan of `keep` over the workgroup
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_sum[wave] = __popcll(m);
    __syncthreads();
    int off = carry_s;
    for (int w = 0; w < wave; ++w) off += wave_sum[w];
    if (keep) collapsed[(size_t)b * n_frames + off + before] = best;
    __syncthreads();
    if (tid == cf - 1) {                               // the chunk's last frame slot (slice 0): total so far, id that precedes the next chunk
      carry_s = off + before + keep;
      last_id_s = (t0 + cf - 1 < n_frames) ? best : last_id_s;
    }
    __syncthreads();
  }
  // entries beyond the row's count: 0 (a valid id), so that the host's table lookup over the whole [B, T'] buffer needs no mask
  for (int i = carry_s + tid; i < n_frames; i += GTH) collapsed[(size_t)b * n_frames + i] = 0;
  if (tid == 0) counts[b] = carry_s;
}

}  // namespace ts

extern "C" int ts_greedy_decode(const float* logits, int32_t batch, int32_t n_classes, int32_t n_frames, int32_t pitch,
                                int32_t* ids, int32_t* collapsed, int32_t* counts, void* stream) {
  if (!logits || !ids || !collapsed || !counts) return TS_EINVAL;
  if (batch <= 0 || n_classes <= 0 || n_frames <= 0 || pitch < n_frames) return TS_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::greedy_kernel, dim3(batch), dim3(ts::GTH), 0, (hipStream_t)stream, logits, n_classes, n_frames,
                     pitch, ids, collapsed, counts);
  return ts::hip_status(hipGetLastError());
}
