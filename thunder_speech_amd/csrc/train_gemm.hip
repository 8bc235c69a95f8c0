// Weight-side kernels of the mixed-precision (bf16) training path's 1x1 convolutions (fine-tuning phase 2, SURVEY 8 config C4;
// quartznet/blocks.py:169-182 with kernel_size = 1) on activation rows [B][C][pitch] (time contiguous):
//
//   forward    v[b][co][t]  = sum_ci W[co][ci] u[b][ci][t]     } both run on the INFERENCE kernel's pointwise-only mode
//   data grad  du[b][ci][t] = sum_co W[co][ci] dv[b][co][t]    } (ts_tcs_subblock_fwd, tcs_kernel.hip: 520-750 TF/s on these shapes,
//                                                                 rocBLAS' strided-batched call reaches 320); what they need from
//                                                                 here is the weights in MFMA B-fragment order, W and W^T:
//                                                                 pack_pw_multi_kernel, one launch for all layers after the optimizer step
//   weight grad dW[co][ci] += sum_{b,t} dv[b][co][t] u[b][ci][t]  wgrad_gemm_kernel + wgrad_reduce_kernel
//
// The weight gradient contracts over (clip, frame): both operands are K-contiguous rows, staged as [128 rows][64 frames] (16-byte
// chunks swizzled by row), fragments by ds_read_b128, v_mfma_f32_32x32x16_bf16, 128 x 128 tiles, 4 waves (2 x 2).  rocBLAS needs B
// clip-sized products plus a reduction of the B partials (25-33 + 10 us per layer); here clip groups are split over workgroups
// until every CU has a tile, so the reduction reads 8-16 partials.
#include "ts_common.hpp"

namespace ts {

typedef unsigned short bf16_t;
constexpr int GK = 64;                 // K per step
constexpr int GTILE = 128;             // tile edge (frames / channels)
constexpr int AROWB = GTILE * 2;       // bytes per LDS row of an activation / transposed-weight tile
constexpr int ATILEB = GK * AROWB;     // 16 KiB

// Workgroups are dealt to the 8 XCDs round-robin by id; tiles that share operands should share an XCD's L2.  Logical tile index of
// workgroup `id`: XCD x = id % 8 owns the contiguous range [x * n / 8, (x + 1) * n / 8) (n a multiple of 8; identity otherwise).
__device__ __forceinline__ int xcd_tile(int id, int n) { return (n & 7) ? id : (id & 7) * (n >> 3) + (id >> 3); }

__device__ __forceinline__ int taddr(int c, int t) { return c * AROWB + ((((t >> 3) ^ ((c & 3) * 5))) << 4) + ((t & 7) << 1); }

struct WgradArgs {
  const bf16_t* dv;       // [B][M][pitch_v]
  const bf16_t* u;        // [B][N][pitch_u]
  float* part;            // [split][M][N] partial products, one per clip group
  const int* len_u;       // null, or per clip: frames >= len_u[b] of u count as zero (the input mask of a MaskedConv1d, applied here)
  int batch, M, N, t, pitch_v, pitch_u, n_mt, n_nt, clips_per_wg;
};

constexpr int WROWB = GK * 2;            // 128-byte rows: 64 frames
constexpr int WTILEB = GTILE * WROWB;    // 16 KiB
__device__ __forceinline__ int waddr(int r, int c) { return r * WROWB + ((c ^ (r & 7)) << 4); }

template <int N>
__device__ __forceinline__ void vm_wait() {
  constexpr int n = N < 63 ? N : 63;
  __builtin_amdgcn_s_waitcnt(0x0F70 | (n & 15) | ((n >> 4) << 14));
  asm volatile("" ::: "memory");
}

// LDS ring depth (stages of one 64-frame K-step: 16 KiB of dv rows + 16 KiB of u rows).  4 for a layer launched on its own (one workgroup per CU:
// three K-steps of loads in flight hide the latency); 2 for the grouped launch -- 64 KiB per workgroup, TWO workgroups per CU, one's loads, prologue and
// 64-KiB result burst under the other's products: 425 -> 358 us per group of 32 layers
constexpr int WD_SINGLE = 4, WD_GROUP = 2;
constexpr int WSTAGEB = 2 * WTILEB;

// Split-K over clip groups: workgroup (tile, sp) contracts clips [sp * clips_per_wg, ...) and stores its f32 partial tile to
// part[sp] (plain coalesced stores; device-scope float atomics cost 18 us per layer here, measured) -- wgrad_reduce_kernel sums
// the partials.  Operand rows arrive by LDS-DMA (buffer_load ... lds, 1 KiB per wave and instruction) into a ring of WD stages, so
// three K-steps of loads are in flight while one is multiplied: a K-step is only 512 MFMA cycles, far less than the load latency.
// The swizzle is applied on the SOURCE address (the DMA's LDS image is lane-linear).  The pitch padding of the last K-step of a clip
// (frames >= T: arbitrary bits) is zeroed in LDS by the wave that fetched those rows, between its own vmcnt wait and the barrier.
template <int WD>
__device__ __forceinline__ void wgrad_tile(const WgradArgs& a, int block_id, int n_blocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int id = xcd_tile(block_id, n_blocks);                 // the tiles of one clip group read the same rows of dv and u
  const int nt_i = id % a.n_nt; id /= a.n_nt;
  const int mt_i = id % a.n_mt;
  const int sp = id / a.n_mt;
  const int m0 = mt_i * GTILE, n0 = nt_i * GTILE;
  const int b_lo = sp * a.clips_per_wg, b_hi = b_lo + a.clips_per_wg < a.batch ? b_lo + a.clips_per_wg : a.batch;
  const int nk = (a.t + GK - 1) / GK, S = (b_hi - b_lo) * nk;
  const i32x4 ra = raw_rsrc(a.dv, (unsigned)((size_t)a.batch * a.M * a.pitch_v * 2));
  const i32x4 rb = raw_rsrc(a.u, (unsigned)((size_t)a.batch * a.N * a.pitch_u * 2));
  // this wave fetches rows [32 wave, 32 wave + 32) of both tiles: DMA q covers 8 rows, lane -> (row lane >> 3, LDS slot lane & 7)
  const int lr = lane >> 3, csrc = (lane & 7) ^ lr;      // logical chunk that lands in this lane's slot (row & 7 == lr)
  int offa[4], offb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 32 * wave + 8 * q + lr;
    const int m = m0 + r < a.M ? m0 + r : a.M - 1, n = n0 + r < a.N ? n0 + r : a.N - 1;
    offa[q] = (m * a.pitch_v + 8 * csrc) * 2;
    offb[q] = (n * a.pitch_u + 8 * csrc) * 2;
  }
  auto issue = [&](int s) {
    const int b = b_lo + s / nk, tk = (s % nk) * GK;
    char* const st = smem + (s % WD) * WSTAGEB;
    const int sa = (b * a.M * a.pitch_v + tk) * 2, sb = (b * a.N * a.pitch_u + tk) * 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      lds_dma16(ra, st + (32 * wave + 8 * q) * WROWB, offa[q], sa);
      lds_dma16(rb, st + WTILEB + (32 * wave + 8 * q) * WROWB, offb[q], sb);
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int s = 0; s < WD - 1 && s < S; ++s) issue(s);
  const int fr = lane & 31, fh = lane >> 5;
  for (int s = 0; s < S; ++s) {
    const int ahead = S - 1 - s < WD - 2 ? S - 1 - s : WD - 2;          // stages issued after s and still allowed in flight
    if (ahead >= 2) vm_wait<16>(); else if (ahead == 1) vm_wait<8>(); else vm_wait<0>();
    char* const st = smem + (s % WD) * WSTAGEB;
    int nv = a.t - (s % nk) * GK;                          // valid frames of this K-step
    if (a.len_u && lane >= 32) {                           // u rows: the clip's own length
      int l = a.len_u[b_lo + s / nk];
      l = l < 0 ? 0 : (l > a.t ? a.t : l);
      nv = l - (s % nk) * GK;
    }
    if (nv < GK) {
      // zero the padding this wave fetched: lane -> one row (lanes 0..31 dv rows, 32..63 u rows), chunks from nv / 8 on
      char* const row = st + (lane >> 5) * WTILEB;
      const int r = 32 * wave + (lane & 31);
      for (int c = nv > 0 ? nv >> 3 : 0; c < 8; ++c) {
        u32x4* const p = reinterpret_cast<u32x4*>(row + waddr(r, c));
        *p = keep_first(*p, nv - 8 * c);
      }
    }
    __builtin_amdgcn_s_barrier();                          // stage s complete in LDS; every wave is done with stage s - 1
    asm volatile("" ::: "memory");
    if (s + WD - 1 < S) issue(s + WD - 1);                 // into the slot of stage s - 1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s16x8 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const s16x8*>(st + waddr(wm * 64 + 32 * i + fr, 2 * ks + fh));
        bf[i] = *reinterpret_cast<const s16x8*>(st + WTILEB + waddr(wn * 64 + 32 * i + fr, 2 * ks + fh));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
  float* const out = a.part + (size_t)sp * a.M * a.N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + 32 * j + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (m < a.M && n < a.N) out[(size_t)m * a.N + n] = acc[i][j][r];
      }
    }
}

__global__ __launch_bounds__(256) void wgrad_gemm_kernel(WgradArgs a) { wgrad_tile<WD_SINGLE>(a, blockIdx.x, gridDim.x); }

// the same product (the weight gradient of the 1x1 MaskedConv1d of a block repeat, reference quartznet/blocks.py:169-182 with kernel_size = 1, as autograd
// computes it) for up to WM_MAX layers in ONE launch: workgroup -> (layer, tile) through the layers' tile-count prefix sums.  A graphed
// training step hands over every layer of a backward piece at once (the weight gradients are off the critical path: nothing reads them before
// the optimizer), so 93 launches of 17-20 us, each with its own launch latency, ramp and drain, become three.
constexpr int WM_MAX = 32;
struct WgradMulti {
  const bf16_t* dv[WM_MAX]; const bf16_t* u[WM_MAX]; float* part[WM_MAX]; const int* len_u[WM_MAX];
  int batch[WM_MAX], M[WM_MAX], N[WM_MAX], t[WM_MAX], pitch_v[WM_MAX], pitch_u[WM_MAX], n_mt[WM_MAX], n_nt[WM_MAX], cpw[WM_MAX];
  int first[WM_MAX + 1];                                  // first workgroup of layer e; first[count] = grid size
  int count;
};
__global__ __launch_bounds__(256) void wgrad_multi_kernel(const WgradMulti m) {
  int e = 0;
  for (int i = 1; i < m.count; ++i) e = (int)blockIdx.x >= m.first[i] ? i : e;
  WgradArgs a;
  a.dv = m.dv[e]; a.u = m.u[e]; a.part = m.part[e]; a.len_u = m.len_u[e];
  a.batch = m.batch[e]; a.M = m.M[e]; a.N = m.N[e]; a.t = m.t[e]; a.pitch_v = m.pitch_v[e]; a.pitch_u = m.pitch_u[e];
  a.n_mt = m.n_mt[e]; a.n_nt = m.n_nt[e]; a.clips_per_wg = m.cpw[e];
  wgrad_tile<WD_GROUP>(a, (int)blockIdx.x - m.first[e], m.first[e + 1] - m.first[e]);
}

// dw[i] += sum_p part[p][i]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long long n, int n_parts) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n) {
    f32x4 s = *reinterpret_cast<const f32x4*>(dw + i);
    for (int p = 0; p < n_parts; ++p) s += *reinterpret_cast<const f32x4*>(part + (size_t)p * n + i);
    *reinterpret_cast<f32x4*>(dw + i) = s;
  } else {
    for (long long k = i; k < n; ++k) {
      float s = dw[k];
      for (int p = 0; p < n_parts; ++p) s += part[(size_t)p * n + k];
      dw[k] = s;
    }
  }
}

// the same for up to RM_MAX layers in one launch (blockIdx.y = layer): a graphed training step parks every layer's partials until the end of
// its backward piece and sums them all at once -- 93 launches of ~8 us (launch-bound: 1-4 MB each) become one or two
constexpr int RM_MAX = 64;
struct ReduceMulti {
  const float* part[RM_MAX];
  float* dw[RM_MAX];
  int n[RM_MAX];
  int n_parts[RM_MAX];
};
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const ReduceMulti a) {
  const int e = blockIdx.y;
  const long long n = a.n[e];
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const float* const part = a.part[e];
  float* const dw = a.dw[e];
  const int n_parts = a.n_parts[e];
  if (i + 4 <= n) {
    f32x4 s = *reinterpret_cast<const f32x4*>(dw + i);
    for (int p = 0; p < n_parts; ++p) s += *reinterpret_cast<const f32x4*>(part + (size_t)p * n + i);
    *reinterpret_cast<f32x4*>(dw + i) = s;
  } else {
    for (long long k = i; k < n; ++k) {
      float s = dw[k];
      for (int p = 0; p < n_parts; ++p) s += part[(size_t)p * n + k];
      dw[k] = s;
    }
  }
}

// B-fragments of v_mfma_f32_32x32x16_bf16 for D[t][n] += X[k][t] * Wn[n][k] (tcs_kernel.hip, plan.pack_pw_frags): group (nt, ks, lane)
// = 8 bf16: Wn[32 nt + lane % 32][16 ks + 8 (lane / 32) + 0..7], groups ordered [n_pad32 / 32][k_pad64 / 16][64].  Per layer two sets:
// the forward one (Wn = W: n = c_out, k = c_in) and the backward one (Wn = W^T: n = c_in, k = c_out).  table rows: (W f32 ptr,
// forward fragments ptr, backward fragments ptr, c_out, c_in) as 64-bit words; blockIdx.y = layer.
__global__ __launch_bounds__(256) void pack_pw_multi_kernel(const unsigned long long* __restrict__ table) {
  const unsigned long long* e = table + (size_t)blockIdx.y * 5;
  const float* w = reinterpret_cast<const float*>(e[0]);
  const int co = (int)e[3], ci = (int)e[4];
  const long long gf = (long long)(round_up(co, 32) / 32) * (round_up(ci, 64) / 16) * 64;
  const long long gb = (long long)(round_up(ci, 32) / 32) * (round_up(co, 64) / 16) * 64;
  long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= gf + gb) return;
  const bool bwd = g >= gf;
  if (bwd) g -= gf;
  const int kt = bwd ? round_up(co, 64) / 16 : round_up(ci, 64) / 16;
  const int lane = (int)(g & 63), ks = (int)((g >> 6) % kt), nt = (int)((g >> 6) / kt);
  const int n = 32 * nt + (lane & 31), k = 16 * ks + 8 * (lane >> 5);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = bwd ? k + j : n, col = bwd ? n : k + j;           // element of W[c_out][c_in]
    v[j] = (row < co && col < ci) ? w[(size_t)row * ci + col] : 0.f;
  }
  unsigned short* const out = reinterpret_cast<unsigned short*>(bwd ? e[2] : e[1]) + (size_t)g * 8;
  *reinterpret_cast<u32x4*>(out) = u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
}

}  // namespace ts

using namespace ts;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int wgrad_split(int batch, int c_in, int c_out) {
  const int tiles = ((c_out + GTILE - 1) / GTILE) * ((c_in + GTILE - 1) / GTILE);
  int split = (cu_count() + tiles - 1) / tiles;            // one workgroup per CU (128 KiB of LDS each)
  split = split < 1 ? 1 : (split > batch ? batch : split);
  const int cpw = (batch + split - 1) / split;
  return (batch + cpw - 1) / cpw;
}

// the grouped launch fills the chip with the tiles of MANY layers, so a layer needs few clip groups: a quarter of the partial tiles to write and
// to sum, four times the K-steps per workgroup behind one prologue and one 64-KiB result burst
static int wgrad_split_group(int batch, int c_in, int c_out) {
  const int s = wgrad_split(batch, c_in, c_out);
  return s < 4 ? s : 4;
}

/* partial [c_out][c_in] tiles per layer that ts_train_pwconv_wgrad_multi leaves in its workspace (= n_parts of ts_train_wgrad_reduce_multi) */
extern "C" int32_t ts_train_pwconv_wgrad_multi_parts(int32_t batch, int32_t c_in, int32_t c_out) {
  if (batch <= 0 || c_in <= 0 || c_out <= 0) return TS_EINVAL;
  const int split = wgrad_split_group(batch, c_in, c_out);
  const int cpw = (batch + split - 1) / split;
  return (batch + cpw - 1) / cpw;
}

/* floats of workspace ts_train_pwconv_wgrad_mfma needs: one [c_out][c_in] partial per clip group */
extern "C" int64_t ts_train_pwconv_wgrad_workspace(int32_t batch, int32_t c_in, int32_t c_out) {
  if (batch <= 0 || c_in <= 0 || c_out <= 0) return TS_EINVAL;
  return (int64_t)wgrad_split(batch, c_in, c_out) * c_in * c_out;
}

/* dw += sum_b dv[b] . u[b]^T; see include/thunder_speech_amd.h */
extern "C" int ts_train_pwconv_wgrad_mfma(const void* dv, const void* u, const int32_t* len_u, float* dw, float* workspace, int32_t batch, int32_t c_in,
                                          int32_t c_out, int32_t t, int32_t pitch_u, int32_t pitch_v, void* stream_) {
  if (!dv || !u || !workspace || batch <= 0 || c_in <= 0 || c_out <= 0 || t <= 0 || pitch_u < t || pitch_v < t) return TS_EINVAL;
  const int tk_end = round_up(t, GK);
  if (c_out % 8 || c_in % 8 || pitch_u % 8 || pitch_v % 8 || pitch_u < tk_end || pitch_v < tk_end || !aligned16(u) || !aligned16(dv) ||
      (dw && !aligned16(dw)) || !aligned16(workspace) || ((size_t)c_out * c_in) % 4)
    return TS_EUNSUPPORTED;
  if ((size_t)batch * c_out * pitch_v * 2 >= (1ull << 31) || (size_t)batch * c_in * pitch_u * 2 >= (1ull << 31)) return TS_EUNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  static bool attr[64] = {};                              // per device (one process may drive several GPUs)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TS_EINVAL;
  if (!attr[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WD_SINGLE * WSTAGEB) != hipSuccess)
      return TS_EUNSUPPORTED;
    attr[dev] = true;
  }
  WgradArgs g;
  g.dv = (const bf16_t*)dv; g.u = (const bf16_t*)u; g.part = workspace; g.len_u = len_u;
  g.batch = batch; g.M = c_out; g.N = c_in; g.t = t; g.pitch_v = pitch_v; g.pitch_u = pitch_u;
  g.n_mt = (c_out + GTILE - 1) / GTILE; g.n_nt = (c_in + GTILE - 1) / GTILE;
  const int split = wgrad_split(batch, c_in, c_out);
  g.clips_per_wg = (batch + split - 1) / split;
  (void)hipGetLastError();
  hipLaunchKernelGGL(wgrad_gemm_kernel, dim3((unsigned)(g.n_mt * g.n_nt * split)), dim3(256), WD_SINGLE * WSTAGEB, stream, g);
  const long long n = (long long)c_out * c_in;
  // dw == NULL: partials only -- the caller sums them later, many layers at once (ts_train_wgrad_reduce_multi)
  if (dw) hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, workspace, dw, n, split);
  return hip_status(hipGetLastError());
}

static int wgrad_check(const void* dv, const void* u, const float* workspace, int batch, int c_in, int c_out, int t, int pitch_u, int pitch_v) {
  if (!dv || !u || !workspace || batch <= 0 || c_in <= 0 || c_out <= 0 || t <= 0 || pitch_u < t || pitch_v < t) return TS_EINVAL;
  const int tk_end = round_up(t, GK);
  if (c_out % 8 || c_in % 8 || pitch_u % 8 || pitch_v % 8 || pitch_u < tk_end || pitch_v < tk_end || !aligned16(u) || !aligned16(dv) ||
      !aligned16(workspace) || ((size_t)c_out * c_in) % 4)
    return TS_EUNSUPPORTED;
  if ((size_t)batch * c_out * pitch_v * 2 >= (1ull << 31) || (size_t)batch * c_in * pitch_u * 2 >= (1ull << 31)) return TS_EUNSUPPORTED;
  return TS_OK;
}

/* the partial products (ts_train_pwconv_wgrad_mfma with dw = NULL) of `count` layers in ceil(count / 32) launches; see include/thunder_speech_amd.h */
extern "C" int ts_train_pwconv_wgrad_multi(const ts_wgrad_item* items, int32_t count, void* stream_) {
  if (!items || count <= 0) return TS_EINVAL;
  for (int e = 0; e < count; ++e) {
    const ts_wgrad_item& it = items[e];
    if (int st = wgrad_check(it.dv, it.u, it.workspace, it.batch, it.c_in, it.c_out, it.t, it.pitch_u, it.pitch_v)) return st;
  }
  hipStream_t stream = (hipStream_t)stream_;
  static bool attr[64] = {};                              // per device (one process may drive several GPUs)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TS_EINVAL;
  if (!attr[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WD_GROUP * WSTAGEB) != hipSuccess)
      return TS_EUNSUPPORTED;
    attr[dev] = true;
  }
  (void)hipGetLastError();
  for (int base = 0; base < count; base += WM_MAX) {
    WgradMulti m;
    m.count = count - base < WM_MAX ? count - base : WM_MAX;
    int grid = 0;
    for (int e = 0; e < WM_MAX; ++e) {
      const ts_wgrad_item& it = items[base + (e < m.count ? e : 0)];
      m.dv[e] = (const bf16_t*)it.dv; m.u[e] = (const bf16_t*)it.u; m.part[e] = it.workspace; m.len_u[e] = it.len_u;
      m.batch[e] = it.batch; m.M[e] = it.c_out; m.N[e] = it.c_in; m.t[e] = it.t; m.pitch_v[e] = it.pitch_v; m.pitch_u[e] = it.pitch_u;
      m.n_mt[e] = (it.c_out + GTILE - 1) / GTILE; m.n_nt[e] = (it.c_in + GTILE - 1) / GTILE;
      const int split = wgrad_split_group(it.batch, it.c_in, it.c_out);
      m.cpw[e] = (it.batch + split - 1) / split;
      m.first[e] = grid;
      if (e < m.count) grid += m.n_mt[e] * m.n_nt[e] * ((it.batch + m.cpw[e] - 1) / m.cpw[e]);
    }
    m.first[WM_MAX] = grid;
    for (int e = m.count; e <= WM_MAX; ++e) m.first[e] = grid;
    hipLaunchKernelGGL(wgrad_multi_kernel, dim3((unsigned)grid), dim3(256), WD_GROUP * WSTAGEB, stream, m);
  }
  return hip_status(hipGetLastError());
}

/* dw[e][i] += sum_p parts[e][p][i] for `count` layers; see include/thunder_speech_amd.h */
extern "C" int ts_train_wgrad_reduce_multi(const void* const* parts, void* const* dws, const int64_t* n, const int32_t* n_parts, int32_t count,
                                           void* stream_) {
  if (!parts || !dws || !n || !n_parts || count <= 0) return TS_EINVAL;
  for (int e = 0; e < count; ++e)
    if (!parts[e] || !dws[e] || n[e] <= 0 || n[e] >= (1ll << 31) || n_parts[e] <= 0 || !aligned16(parts[e]) || !aligned16(dws[e]) || n[e] % 4) return TS_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();
  for (int base = 0; base < count; base += RM_MAX) {
    ReduceMulti a;
    const int m = count - base < RM_MAX ? count - base : RM_MAX;
    long long max_n = 0;
    for (int e = 0; e < RM_MAX; ++e) {
      const int src = base + (e < m ? e : 0);
      a.part[e] = static_cast<const float*>(parts[src]); a.dw[e] = static_cast<float*>(dws[src]); a.n[e] = (int)n[src]; a.n_parts[e] = n_parts[src];
      if (e < m && n[src] > max_n) max_n = n[src];
    }
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)((max_n / 4 + 255) / 256), (unsigned)m), dim3(256), 0, stream, a);
  }
  return hip_status(hipGetLastError());
}

/* W (f32 [c_out][c_in]) of every listed layer -> bf16 MFMA B-fragments of W and of W^T; see include/thunder_speech_amd.h */
extern "C" int ts_train_pack_pw_multi(const void* table, int32_t n_tensors, int64_t max_groups, void* stream_) {
  if (!table || n_tensors <= 0 || max_groups <= 0) return TS_EINVAL;
  if (n_tensors > 65535) return TS_EUNSUPPORTED;
  (void)hipGetLastError();
  hipLaunchKernelGGL(pack_pw_multi_kernel, dim3((unsigned)((max_groups + 255) / 256), n_tensors), dim3(256), 0, (hipStream_t)stream_,
                     static_cast<const unsigned long long*>(table));
  return hip_status(hipGetLastError());
}
