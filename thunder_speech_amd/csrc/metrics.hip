// Text side of the training / validation step on the device.
//
// Reference: BaseCTCModule.validation_step (module.py:129-163) feeds decoded strings to torchmetrics' CharErrorRate /
// WordErrorRate = sum of Levenshtein distances / sum of reference lengths, computed pair by pair in Python;
// BatchTextTransformer.encode (text_processing/transform.py:65-92) tokenises, adds the start / end tokens, numericalises
// token by token and pads with pad_idx.
//   ts_edit_distance : one workgroup per (prediction, reference) pair of int32 symbol sequences (characters or word ids),
//                      anti-diagonal dynamic programme in LDS, unit costs (insert / delete / substitute).
//   ts_encode_chars  : character-level vocabularies (QuartzNet labels): code points -> ids by binary search in the sorted
//                      vocabulary, optional start / end ids, padded [n][s_max] int64 + lengths int64.
#include "ts_common.hpp"

namespace ts {

__global__ __launch_bounds__(256) void edit_distance_kernel(const int* __restrict__ a, const int* __restrict__ a_off,
                                                             const int* __restrict__ b, const int* __restrict__ b_off,
                                                             int* __restrict__ dist) {
  extern __shared__ int sm[];
  const int p = blockIdx.x;
  const int* sa = a + a_off[p];
  const int* sb = b + b_off[p];
  const int n = a_off[p + 1] - a_off[p], m = b_off[p + 1] - b_off[p];
  if (n == 0 || m == 0) { if (threadIdx.x == 0) dist[p] = n + m; return; }
  // d[i][j], i in [0, n], j in [0, m]; diagonal k = i + j holds the cells with i in [max(0, k - m), min(n, k)], stored by i
  int* d0 = sm;                  // diagonal k - 2
  int* d1 = sm + (n + 1);        // diagonal k - 1
  int* d2 = sm + 2 * (n + 1);    // diagonal k
  for (int k = 0; k <= n + m; ++k) {
    const int lo = k > m ? k - m : 0, hi = k < n ? k : n;
    for (int i = lo + threadIdx.x; i <= hi; i += 256) {
      const int j = k - i;
      int v;
      if (i == 0) v = j;
      else if (j == 0) v = i;
      else {
        const int sub = d0[i - 1] + (sa[i - 1] != sb[j - 1] ? 1 : 0);
        const int del = d1[i - 1] + 1;          // d[i-1][j]
        const int ins = d1[i] + 1;              // d[i][j-1]
        v = min(sub, min(del, ins));
      }
      d2[i] = v;
    }
    __syncthreads();
    int* t = d0; d0 = d1; d1 = d2; d2 = t;
  }
  if (threadIdx.x == 0) dist[p] = d1[n];
}

__global__ __launch_bounds__(256) void encode_chars_kernel(const int* __restrict__ text, const int* __restrict__ off,
                                                            const int* __restrict__ vocab_cp, const int* __restrict__ vocab_id,
                                                            int n_vocab, int unk_id, int start_id, int end_id, int pad_id, int s_max,
                                                            long long* __restrict__ out, long long* __restrict__ lens) {
  const int r = blockIdx.x;
  const int* t = text + off[r];
  const int n = off[r + 1] - off[r];
  const int pre = start_id >= 0 ? 1 : 0, post = end_id >= 0 ? 1 : 0;
  const int total = n + pre + post;
  for (int j = threadIdx.x; j < s_max; j += 256) {
    long long v = pad_id;
    if (j < total) {
      if (pre && j == 0) v = start_id;
      else if (post && j == total - 1) v = end_id;
      else {
        const int cp = t[j - pre];
        int lo = 0, hi = n_vocab - 1, id = unk_id;
        while (lo <= hi) {
          const int mid = (lo + hi) >> 1;
          const int c = vocab_cp[mid];
          if (c == cp) { id = vocab_id[mid]; break; }
          if (c < cp) lo = mid + 1; else hi = mid - 1;
        }
        v = id;
      }
    }
    out[(size_t)r * s_max + j] = v;
  }
  if (threadIdx.x == 0) lens[r] = total;
}

}  // namespace ts

extern "C" int ts_edit_distance(const int32_t* a, const int32_t* a_off, const int32_t* b, const int32_t* b_off, int32_t n_pairs,
                                int32_t max_a_len, int32_t* dist, void* stream) {
  if (!a_off || !b_off || !dist || n_pairs <= 0 || max_a_len < 0) return TS_EINVAL;
  const size_t lds = (size_t)3 * (max_a_len + 1) * sizeof(int);
  if (lds > 64 * 1024) return TS_EUNSUPPORTED;            // sequences beyond ~5400 symbols
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::edit_distance_kernel, dim3(n_pairs), dim3(256), lds, (hipStream_t)stream, a, a_off, b, b_off, dist);
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_encode_chars(const int32_t* text, const int32_t* off, int32_t n_rows, const int32_t* vocab_cp,
                               const int32_t* vocab_id, int32_t n_vocab, int32_t unk_id, int32_t start_id, int32_t end_id,
                               int32_t pad_id, int32_t s_max, int64_t* out, int64_t* lens, void* stream) {
  if (!off || !out || !lens || n_rows <= 0 || s_max <= 0 || n_vocab < 0 || (n_vocab && (!vocab_cp || !vocab_id))) return TS_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::encode_chars_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, text, off, vocab_cp, vocab_id, n_vocab,
                     unk_id, start_id, end_id, pad_id, s_max, reinterpret_cast<long long*>(out), reinterpret_cast<long long*>(lens));
  return ts::hip_status(hipGetLastError());
}
