// Training-time augmentation on the device: SpecAugment / SpecCutout mask geometry + application (dropout: csrc/train_extra.hip).
//
// Reference: quartznet/spec_augment.py:23-102 (SpecAugment = torchaudio.functional.mask_along_axis per mask, SpecCutout =
// _create_mask twice per rectangle), wired into FilterbankFeatures after the normaliser (quartznet/transform.py:299-320);
// nn.Dropout after every ReLU of the encoder blocks (quartznet/blocks.py:227-228) and inside linear_decoder
// (blocks.py:226-248).  The reference draws the mask geometry with torch.rand(1) on the HOST and applies masked_fill passes
// over the whole [B, F, T] tensor; here a mask is a row (f0, f1, t0, t1) of a small device table:
//   * the front end applies the table inside its normalise/transposition kernel (csrc/frontend.hip) -- no extra pass;
//   * ts_spec_mask_apply zeroes the rectangles of an existing feature tensor (standalone module call);
//   * ts_spec_masks_draw fills the table from a Philox stream on the device (graph-capturable); the Python side can also
//     fill it with the reference's own host draws, which reproduces the reference bit for bit under torch.manual_seed.
#include "ts_common.hpp"
#include "ts_philox.hpp"

namespace ts {

// torchaudio.functional.mask_along_axis (0.12) / spec_augment._create_mask: value = rand * mask_param;
// min_value = rand * (size - value); [start, end) = [long(min_value), long(min_value) + long(value))
__host__ __device__ inline void draw_span(float u_value, float u_min, int mask_param, int size, int& start, int& end) {
  const float value = u_value * (float)mask_param;
  const float min_value = u_min * ((float)size - value);
  start = (int)min_value;
  end = (int)min_value + (int)value;
}

struct SpecDrawArgs {
  unsigned long long seed;
  int n_time, time_width, n_freq, freq_width, n_cutout, cut_time_width, cut_freq_width;
  int n_mels, n_frames;
  int* table;      // [n][4] = f0, f1, t0, t1
};

__global__ void spec_draw_kernel(const SpecDrawArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int row = 0;
  unsigned long long ctr = 0;
  auto two = [&](float& u0, float& u1) { const Philox4 r = philox(a.seed, PHILOX_SPEC, ctr++); u0 = u01(r.v[0]); u1 = u01(r.v[1]); };
  // SpecCutout (transform.py:299-308 puts it first): frequency span, then a time span drawn with FREQ_width
  // (spec_augment.py:99-100 passes self.freq_width to both -- reference quirk, kept)
  for (int i = 0; i < a.n_cutout; ++i, ++row) {
    float u0, u1;
    int f0, f1, t0, t1;
    two(u0, u1); draw_span(u0, u1, a.cut_freq_width, a.n_mels, f0, f1);
    two(u0, u1); draw_span(u0, u1, a.cut_freq_width, a.n_frames, t0, t1);
    a.table[4 * row + 0] = f0; a.table[4 * row + 1] = f1; a.table[4 * row + 2] = t0; a.table[4 * row + 3] = t1;
  }
  // SpecAugment: time masks first, then frequency masks (spec_augment.py:51-56)
  for (int i = 0; i < a.n_time; ++i, ++row) {
    float u0, u1;
    int t0, t1;
    two(u0, u1); draw_span(u0, u1, a.time_width, a.n_frames, t0, t1);
    a.table[4 * row + 0] = 0; a.table[4 * row + 1] = a.n_mels; a.table[4 * row + 2] = t0; a.table[4 * row + 3] = t1;
  }
  for (int i = 0; i < a.n_freq; ++i, ++row) {
    float u0, u1;
    int f0, f1;
    two(u0, u1); draw_span(u0, u1, a.freq_width, a.n_mels, f0, f1);
    a.table[4 * row + 0] = f0; a.table[4 * row + 1] = f1; a.table[4 * row + 2] = 0; a.table[4 * row + 3] = a.n_frames;
  }
}

// one workgroup per (mask, clip): zero the rectangle rows [f0, f1) x frames [t0, t1) of a [B][C][pitch] tensor (bf16 or f32)
template <class E>
__global__ __launch_bounds__(256) void spec_apply_kernel(E* __restrict__ x, int channels, int t, int pitch, const int* __restrict__ table) {
  const int* m = table + 4 * blockIdx.x;
  const int f0 = max(m[0], 0), f1 = min(m[1], channels), t0 = max(m[2], 0), t1 = min(m[3], t);
  if (f1 <= f0 || t1 <= t0) return;
  const int w = t1 - t0;
  E* base = x + (size_t)blockIdx.y * channels * pitch;
  for (int idx = threadIdx.x; idx < (f1 - f0) * w; idx += 256) base[(size_t)(f0 + idx / w) * pitch + t0 + idx % w] = E(0);
}

}  // namespace ts

extern "C" int ts_spec_masks_draw(uint64_t seed, int32_t n_time, int32_t time_width, int32_t n_freq, int32_t freq_width,
                                  int32_t n_cutout, int32_t cut_time_width, int32_t cut_freq_width, int32_t n_mels,
                                  int32_t n_frames, int32_t* table, void* stream) {
  if (!table || n_time < 0 || n_freq < 0 || n_cutout < 0 || n_mels <= 0 || n_frames <= 0) return TS_EINVAL;
  if (n_time + n_freq + n_cutout == 0) return TS_OK;
  ts::SpecDrawArgs a{seed, n_time, time_width, n_freq, freq_width, n_cutout, cut_time_width, cut_freq_width, n_mels, n_frames, table};
  (void)hipGetLastError();
  hipLaunchKernelGGL(ts::spec_draw_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
  return ts::hip_status(hipGetLastError());
}

extern "C" int ts_spec_mask_apply(void* features, int32_t elem_bytes, int32_t batch, int32_t channels, int32_t t, int32_t pitch,
                                  const int32_t* table, int32_t n_masks, void* stream) {
  if (!features || batch <= 0 || channels <= 0 || t <= 0 || pitch < t || n_masks < 0 || (n_masks && !table)) return TS_EINVAL;
  if (elem_bytes != 2 && elem_bytes != 4) return TS_EINVAL;
  if (!n_masks) return TS_OK;
  (void)hipGetLastError();
  if (elem_bytes == 2)
    hipLaunchKernelGGL(ts::spec_apply_kernel<unsigned short>, dim3(n_masks, batch), dim3(256), 0, (hipStream_t)stream,
                       static_cast<unsigned short*>(features), channels, t, pitch, table);
  else
    hipLaunchKernelGGL(ts::spec_apply_kernel<float>, dim3(n_masks, batch), dim3(256), 0, (hipStream_t)stream,
                       static_cast<float*>(features), channels, t, pitch, table);
  return ts::hip_status(hipGetLastError());
}
