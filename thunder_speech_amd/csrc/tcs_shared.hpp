// Shared between the fused TCS kernels (csrc/tcs_kernel.hip: first design / generic kernel and the C-ABI dispatch; csrc/tcs_split.hip:
// split kernel).
#pragma once
#include "ts_common.hpp"

#include <type_traits>

namespace ts {

constexpr int KC = 64;         // input channels per stage
constexpr int NKP = 3;         // depthwise k-steps (of 4 samples) per pass

struct TcsArgs {
  const unsigned short* x;     // [B][c_in][pitch_in]
  const unsigned short* xres;  // [B][c_res][pitch_res]
  void* y;                     // [B][c_out][pitch_out] bf16 or f32
  const int* len_in;
  const int* len_res;
  const unsigned short* taps;  // [c_in_pad][4][4*nk]
  const unsigned short* taps_raw;  // split kernel: raw tap image, see plan.pack_dw_taps_raw
  const unsigned short* pw_w;  // fragments (32x32x16 form)
  const unsigned short* res_w;
  const unsigned short* pw_w16;   // fragments of the 16x16x32 form (split kernel)
  const unsigned short* res_w16;
  const float* bias;
  int batch, c_in, c_out, c_res;
  int pitch_in, pitch_out, pitch_res;
  int t_out;
  int kernel, stride, dilation, padding;
  int npass;                   // nk = 3 * npass
  int woff;                    // padL8 - padL4: element offset of the lane windows inside an xs row
  int padl8;                   // xs row starts at input frame t0*stride - padl8
  int xe;                      // staged elements per xs row (multiple of 64)
  int xuse;                    // elements of a row the depthwise actually reads
  int xpitch;                  // xs row pitch in elements (8-byte aligned rows, pitch == 8 mod 16 bytes)
  int relu;
  int res_stride;
  int kt_main, kt_res;         // k-steps (16 channels) in the packed weights = c_pad64 / 16
  int taps_lds;                // 1: taps of the stage are cached in LDS
  int n_tt, n_z, n_tiles;      // tile grid: time tiles, output-channel splits, total
  int zero_tail;               // 1: store 0 for frames >= the output length (keeps the tail-zero invariant)
  int xcd;                     // split kernel: 1 = XCD-contiguous tile order (grid is a multiple of 8)
  const unsigned short* se_y;  // squeeze-excite tail in the epilogue (pointwise-only split launches): see SplitLayer
  const float* se_gate;
  float* stats;                // generic pointwise-only launches: per-tile (sum y, sum y^2) per output channel, f32 [c_out][batch * n_tt][2], or null
};

__device__ __forceinline__ int conv_len(int len, int k, int s, int p, int d) {
  const int num = len + 2 * p - d * (k - 1) - 1;
  return num < 0 ? 0 : num / s + 1;
}

// LDS writes of this wave complete -> workgroup barrier.  Outstanding global loads stay in flight.
__device__ __forceinline__ void stage_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ unsigned relu_bf16x2(unsigned v) {
  // max(x, 0) on two packed bf16: sign-magnitude floats order like signed 16-bit integers around zero
  const s16x2 r = __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), s16x2{0, 0});
  return __builtin_bit_cast(unsigned, r);
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [A, B)
template <int A, int B, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (A < B) {
    f(std::integral_constant<int, A>{});
    static_for<A + 1, B>(f);
  }
}
// at most N vector-memory operations outstanding (N is clamped to the 6-bit counter)
template <int N>
__device__ __forceinline__ void vm_wait() {
  constexpr int n = N < 63 ? N : 63;
  __builtin_amdgcn_s_waitcnt(0x0F70 | (n & 15) | ((n >> 4) << 14));
  asm volatile("" ::: "memory");
}

// position of a tile in the (clip, output-channel split, time tile) grid, advanced by the grid stride without divisions
struct TilePos {
  int b, z, tt;
  int sb, sz, st;       // the stride, decomposed the same way
  __device__ __forceinline__ void init(int tile, int step, int n_tt, int n_z) {
    tt = tile % n_tt; z = (tile / n_tt) % n_z; b = (tile / n_tt) / n_z;
    st = step % n_tt; sz = (step / n_tt) % n_z; sb = (step / n_tt) / n_z;
  }
  // branch-free conditional advance (scalar selects)
  __device__ __forceinline__ void advance_if(bool go, int n_tt, int n_z) {
    tt += go ? st : 0;
    const int c1 = tt >= n_tt ? 1 : 0;
    tt -= c1 ? n_tt : 0;
    z += (go ? sz : 0) + c1;
    const int c2 = z >= n_z ? 1 : 0;
    z -= c2 ? n_z : 0;
    b += (go ? sb : 0) + c2;
  }
  __device__ __forceinline__ void advance(int n_tt, int n_z) {
    tt += st;
    const int c1 = tt >= n_tt ? 1 : 0;
    tt -= c1 ? n_tt : 0;
    z += sz + c1;
    const int c2 = z >= n_z ? 1 : 0;
    z -= c2 ? n_z : 0;
    b += sb + c2;
  }
};


// [ci][t] bf16 tile of the depthwise output / identity input, 16-byte chunks XOR-swizzled so that both
// the 8-byte row writes and the transposed reads spread over the banks.
template <int TT>
struct DwTile {
  static constexpr int ROWB = TT * 2;
  static constexpr int BYTES = KC * ROWB;
  __device__ static __forceinline__ int sw(int c) {
    return TT == 128 ? (c & 3) * 5 : ((((c >> 1) & 1) << 2) | (c & 3));
  }
  __device__ static __forceinline__ int addr(int c, int t) {
    return c * ROWB + ((((t >> 3) ^ sw(c))) << 4) + ((t & 7) << 1);
  }
};

// ---- split kernel (csrc/tcs_split.hip) ------------------------------------------------------------------------------------
struct SplitLayer {
  const unsigned short* x;         // [B][c_in][pitch]
  const unsigned short* xres;      // [B][c_res][pitch_res] residual input, or the input of a pointwise-only layer
  unsigned short* y;               // [B][c_out][pitch]
  const unsigned short* taps_raw;  // raw tap image (plan.pack_dw_taps_raw)
  const unsigned short* pw_w;      // B fragments of v_mfma_f32_16x16x32_bf16: [c_out / 16][c_in / 32][64][8]
  const unsigned short* res_w;
  const float* bias;
  int c_in, c_res, pitch_res, relu;
  int kt_main, kt_res;             // k-steps (16 channels) of the packed weights
  // squeeze-excite tail (SE instantiation only): y = relu(se_gate[b][co] * se_y[b][co][t] + this layer's result) -- the closing step of a
  // CitrinetBlock (citrinet/blocks.py:186-196) in the residual 1x1 launch's epilogue; se_y has the pitch of y
  const unsigned short* se_y;
  const float* se_gate;
};
struct SplitArgs {
  SplitLayer layer;
  const int* len;                  // int32 [B] valid frames
  int batch, c_out, pitch_in, pitch_out, t_out;
  int kernel, padding, dilation;
  int woff, padl8;
  int n_tt, n_z, n_tiles;
  int zero_tail, xcd;
};
// TS_EUNSUPPORTED when no instantiation fits (npass = depthwise passes of 3 k-steps, xe = staged frames per row, a multiple of 64;
// wm = 1: 96-frame x 512-channel tiles, 2: 192 x 256; dil = 1, or 2 for the phase-split form of a dilation-2 layer)
int launch_split_layer(SplitArgs& a, int npass, int xe, int wm, int dil, hipStream_t stream);
// pointwise layer with <= 32 output channels and f32 results (csrc/pw_logits.hip); TS_EUNSUPPORTED when the shape is not its
int launch_pw_logits(const TcsArgs& w, hipStream_t stream);
// time-tile choice of the split kernel for a layer: 1 = 96 frames, 2 = 192 frames (c_out <= 256 only)
int split_tile_wm(int c_out);

}  // namespace ts
