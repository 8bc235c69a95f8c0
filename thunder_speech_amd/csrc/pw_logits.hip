// Pointwise convolution with few output channels and f32 results -- the decoder of the CTC models: logits[b, v, t] = sum_c W[v, c] x[b, c, t] + bias[v]
// (reference blocks.py:199-216 conv1d_decoder: nn.Conv1d(C, V, kernel_size=1); QuartzNet: 1024 -> 29 on 64 x 751 frames).  98 MB of bf16 input for
// 5.6 MB of logits: a pure read stream.  The generic sub-block kernel (4 producer + 4 consumer waves, one barrier per 64-channel stage) took 44.6 us
// for it; this kernel has no workgroup-level hand-over inside the contraction:
//   workgroup = (clip, 96- or 128-frame tile), 4 waves; wave w contracts channels [w C/4, (w + 1) C/4) in k-steps of 32 channels:
//     rows global -> registers (16 B per lane, two k-steps ahead) -> wave-private LDS tile [32 ch][<= 128 frames] (XOR-swizzled 16-byte chunks, the
//     split kernel's layout) -> ds_read_b64_tr_b16 A fragments (frames x channels) x pre-packed weight fragments (pw_w16) on
//     v_mfma_f32_16x16x32_bf16 -> D[frames][32 outputs] in 48 / 64 accumulator registers;
//   the four partial sums meet once, at the end, through the (then idle) staging tiles, half of the tile at a time.
// Tile length by balance: two workgroups fit a CU, so 64 x 751 frames are 512 tiles of 96 frames (exactly two per CU) rather than 384 of 128
// (half the CUs would run two tiles, the others one: 30.7 us measured; the same reasoning as the split kernel's 96-frame granules).
// Input frames >= len[b] count as 0 (the masking every sub-block launch applies); `zero_tail` zeroes the results from the length on.
#include "tcs_shared.hpp"

namespace ts {

namespace {

constexpr int LK = 32;                        // channels per k-step
constexpr int LROWB = 256;                    // bytes per staged channel row

struct LogitArgs {
  const unsigned short* x; const unsigned short* w16; const float* bias; float* y; const int* len;
  int batch, c_in, c_out, pitch_in, pitch_out, t_out, relu, zero_tail, n_tt;
};

template <int MT>                             // 16-frame accumulator tiles per workgroup tile: 6 (96 frames) or 8 (128)
__global__ __launch_bounds__(256, 2) void pw_logits_kernel(const LogitArgs a) {
  constexpr int LT = 16 * MT, NCH = LT / 8, MH = MT / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];           // [4 waves][LK][LROWB]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / a.n_tt, t0 = (blockIdx.x % a.n_tt) * LT;
  char* const tile = smem + wave * (LK * LROWB);
  auto key = [](int c) { return ((c & 3) * 5) ^ (((c >> 3) & 1) << 1); };
  const int len_b = a.len ? min(max(a.len[b], 0), a.t_out) : a.t_out;
  const int c_w = a.c_in / 4, c0 = wave * c_w, n_ks = c_w / LK;
  // staging: instruction i of a k-step covers rows 4 i + (lane >> 4), 16-byte chunk lane & 15
  const int srow = lane >> 4, sch = lane & 15;
  const unsigned short* const xg = a.x + ((size_t)b * a.c_in + c0 + srow) * a.pitch_in + t0 + 8 * sch;
  const int nkeep = len_b - (t0 + 8 * sch);                             // valid frames of this lane's chunk (may be <= 0 or >= 8)
  int swr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) swr[i] = (4 * i + srow) * LROWB + ((sch ^ key(4 * i + srow)) << 4);
  // transposed reads (the split kernel's consumer form): lane group kg reads channels 8 kg + q4 (+ 4), frames 16 mt + 4 p4 ..
  const int kg = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int c = 8 * kg + q4, t = 16 * mt + 4 * p4;
    abase[mt] = c * LROWB + (((t >> 3) ^ key(c)) << 4) + ((t & 7) << 1);
  }
  const int kt = a.c_in / LK;                                           // k-steps of the packed weights per 16-output tile
  const unsigned short* const wg = a.w16 + ((size_t)(c0 / LK) * 64 + lane) * 8;

  f32x4 acc[MT][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int co = 16 * nt + (lane & 15);
    const float bv = (wave == 0 && co < a.c_out) ? a.bias[co] : 0.f;   // the bias enters once, as wave 0's initial accumulator
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = f32x4{bv, bv, bv, bv};
  }
  u32x4 X[2][8];
  auto fetch = [&](u32x4 (&R)[8], int ks) {
    const unsigned short* p = xg + (size_t)ks * LK * a.pitch_in;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (sch < NCH) R[i] = *reinterpret_cast<const u32x4*>(p + (size_t)4 * i * a.pitch_in);
  };
  s16x8 wnext[2];                                                       // weight fragments of the NEXT k-step (an L2 round trip ahead of their use)
  auto load_w = [&](int ks) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) wnext[nt] = *reinterpret_cast<const s16x8*>(wg + ((size_t)nt * kt + ks) * 512);
  };
  auto step = [&](u32x4 (&R)[8], int ks) {
    const s16x8 wf[2] = {wnext[0], wnext[1]};
    if (ks + 1 < n_ks) load_w(ks + 1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (sch < NCH) *reinterpret_cast<u32x4*>(tile + swr[i]) = nkeep >= 8 ? R[i] : keep_first(R[i], nkeep);
    if (ks + 2 < n_ks) fetch(R, ks + 2);                                // the set just consumed takes the k-step after next
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)tile + abase[mt]));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TS_LDS s16x4*)((TS_LDS char*)tile + abase[mt] + 4 * LROWB));
      const s16x8 af = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf[nt], acc[mt][nt], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // the tile is rewritten by the next step
  };
  load_w(0);
  fetch(X[0], 0);
  if (n_ks > 1) fetch(X[1], 1);
  for (int ks = 0; ks < n_ks; ks += 2) {
    step(X[0], ks);
    if (ks + 1 < n_ks) step(X[1], ks + 1);
  }
  // ---- the four partial sums meet in the staging tiles, half of the frames at a time: [wave][mt' < MH][nt][i][lane] f32 <= 8 KiB per wave
  float* const mine = reinterpret_cast<float*>(tile);
  const float* const all = reinterpret_cast<const float*>(smem);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();                                                     // every wave is done with its tile (h = 0) / with the previous half's sums
#pragma unroll
    for (int m = 0; m < MH; ++m)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) mine[((m * 2 + nt) * 4 + i) * 64 + lane] = acc[MH * h + m][nt][i];
    __syncthreads();
    if (wave >= MH) continue;                                            // MH = 3: the fourth wave has nothing to finish
    const int mt = MH * h + wave;                                        // this wave finishes frames [16 mt, 16 mt + 16) of the tile
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] += all[w * (LK * LROWB / 4) + ((wave * 2 + nt) * 4 + i) * 64 + lane];
      const int co = 16 * nt + (lane & 15), t = t0 + 16 * mt + 4 * kg;
      if (co >= a.c_out || t >= a.t_out) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (a.relu) s[i] = s[i] > 0.f ? s[i] : 0.f;
        if (a.zero_tail && t + i >= len_b) s[i] = 0.f;
      }
      float* const dst = a.y + ((size_t)b * a.c_out + co) * a.pitch_out + t;
      if (t + 4 <= a.t_out) {
        *reinterpret_cast<f32x4*>(dst) = s;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (t + i < a.t_out) dst[i] = s[i];
      }
    }
  }
}

}  // namespace

// TS_EUNSUPPORTED unless: pointwise, f32 result, no residual, at most 32 output channels, c_in a multiple of 128, 16x16x32 weight fragments given,
// rows of x readable up to the end of the last 128-frame tile
int launch_pw_logits(const TcsArgs& w, hipStream_t stream) {
  if (!w.pw_w16 || w.c_res > 0 || w.c_out > 32 || w.c_in % (4 * LK) || w.stride != 1 || w.kernel != 1) return TS_EUNSUPPORTED;
  // two workgroups per CU: the tile length that needs fewer frames per CU over the whole launch (ties: the longer tile)
  const long long slots = 2ll * cu_count();
  const long long n96 = (long long)w.batch * ((w.t_out + 95) / 96), n128 = (long long)w.batch * ((w.t_out + 127) / 128);
  const bool t96 = ((n96 + slots - 1) / slots) * 96 < ((n128 + slots - 1) / slots) * 128;
  const int lt = t96 ? 96 : 128;
  const int n_tt = (w.t_out + lt - 1) / lt;
  if (w.pitch_in < n_tt * lt || w.pitch_out % 4 || reinterpret_cast<uintptr_t>(w.y) % 16 || reinterpret_cast<uintptr_t>(w.x) % 16) return TS_EUNSUPPORTED;
  LogitArgs a{w.x, w.pw_w16, w.bias, static_cast<float*>(w.y), w.len_in, w.batch, w.c_in, w.c_out, w.pitch_in, w.pitch_out, w.t_out, w.relu, w.zero_tail, n_tt};
  const size_t lds = 4 * LK * LROWB;
  (void)hipGetLastError();
  if (t96) hipLaunchKernelGGL(pw_logits_kernel<6>, dim3((unsigned)(w.batch * n_tt)), dim3(256), lds, stream, a);
  else hipLaunchKernelGGL(pw_logits_kernel<8>, dim3((unsigned)(w.batch * n_tt)), dim3(256), lds, stream, a);
  return hip_status(hipGetLastError());
}

}  // namespace ts
