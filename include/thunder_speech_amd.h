/*
 * thunder_speech_amd.h -- C ABI of the MI355X-native (gfx950) hot path of thunder-speech.
 *
 * The reference (scart97/thunder-speech 3.2.0) has no FFI: its hot path is a chain of ATen ops called
 * from Python.  Each entry point below replaces one such call site (cited as file:line of
 * /root/reference) with a hand-written HIP kernel.  All pointers are DEVICE pointers into buffers
 * owned by the caller (PyTorch-ROCm allocates them); `stream` is a hipStream_t passed as void*
 * (NULL = the default stream).  Every function returns 0 on success, a negative TS_E* code on an
 * argument error, or a positive hipError_t value when a HIP call failed.  Nothing here allocates,
 * frees or synchronises, so every call can be captured into a hipGraph.
 *
 * Activation layout ("NCT-p"): [B][C][Tp] bf16, time contiguous, Tp = ts_time_pitch(T) = round_up(T + 384, 128);
 * the reference's own [B, C, T] layout with a padded pitch.  Columns >= T are scratch (or zero, see
 * TS_TCS_IN_TAILZERO).
 */
#ifndef THUNDER_SPEECH_AMD_H
#define THUNDER_SPEECH_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TS_OK 0
#define TS_EINVAL (-1)       /* bad argument / unsupported shape */
#define TS_EUNSUPPORTED (-2) /* valid reference configuration this build has no kernel for */

#define TS_ABI_VERSION 11

/* Library identification: ABI version and the gfx target the code objects were built for. */
int ts_abi_version(void);
const char* ts_build_target(void);

/* Time pitch (elements) used for an activation with T frames. */
int ts_time_pitch(int T);
/* frames per tile of the masked pointwise-only launch for this shape (see ts_tcs_desc.stats) */
int ts_tcs_pointwise_tile_frames(int32_t batch, int32_t c_out, int32_t t_out);
/* Tile shape of the tail-zero pointwise-only launches with more than 256 output channels (Citrinet's residual 1x1 convs, QuartzNet's last block):
 * 0 = 96 frames x 512 channels per workgroup, consumer waves of 96 x 64; 1 = 192 x 256 per workgroup, consumer waves of 192 frames x 32
 * channels (half the weight-fragment loads per matrix instruction).  Same results bit for bit.  Process-wide; returns the previous mode. */
int ts_tcs_pointwise_wide(int32_t on);

/* ------------------------------------------------------------------------------------------------
 * Fused time-channel-separable sub-block (inference):
 *   y = act( pointwise( mask(depthwise_K( mask(x) )) ) * bn_scale + bn_shift  [+ residual 1x1(mask(x_res))] )
 * replaces, per sub-block, the ATen chain  masked_fill -> conv1d(groups=C) -> masked_fill -> conv1d(k=1)
 * -> batch_norm -> [add] -> relu   (quartznet/blocks.py:166-182, :195-224, :317-338;
 * citrinet/blocks.py:175-197).  BN (eval) is folded into `pw_w`/`bias` by the caller.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ts_tcs_desc {
  /* geometry */
  int32_t batch;
  int32_t c_in, c_out;          /* true channel counts */
  int32_t t_in, t_out;          /* frames in the input / output tensors (t_out = conv output size) */
  int32_t pitch_in, pitch_out;  /* time pitches of x and y (elements) */
  int32_t kernel, stride, dilation, padding;
  int32_t depthwise;            /* 1: depthwise(K) then pointwise; 0: pointwise only (kernel must be 1) */
  int32_t relu;                 /* apply ReLU in the epilogue */
  int32_t out_fp32;             /* 1: y is float [B][c_out][pitch_out] (decoder logits), 0: bf16 */
  /* residual branch (1x1 conv of the block input, accumulated before the activation) */
  int32_t c_res;                /* 0 = none */
  int32_t pitch_res, t_res, res_stride;
  /* prepacked parameters (see thunder_speech_amd/plan.py for the packers) */
  int32_t dw_ksteps;            /* NK: number of 4-sample k-steps in `dw_taps` (multiple of 3) */
  int32_t flags;                /* TS_TCS_* bits below */
  const void* dw_taps;          /* bf16 [c_in_pad64/64][4 waves][NK][64 lanes][4]  shifted Toeplitz rows of the depthwise taps,
                                   lane = 4*(channel % 16) + row (plan.tap_fragments) */
  const void* dw_taps_raw;      /* stride-1 layers (may be NULL: the kernels that read the pre-shifted fragments run instead): the same
                                   NK k-steps as raw taps, bf16 [c_in_pad64/64][4 groups of 16 channels][KiB-padded image]; per channel
                                   16 NK + 16 bytes = two copies of wp[n] = w[n - 3 - (round_up(padding, 4) - padding)], n < 4 NK + 4, the
                                   second shifted by one element (plan.pack_dw_taps_raw).  With TS_TCS_TAPS_PHASE: the phase-split form. */
  const void* pw_w;             /* bf16 [c_out_pad32/32][c_in_pad64/16][64][8]  MFMA B-fragments of W*bn_scale */
  const void* res_w;            /* bf16 [c_out_pad32/32][c_res_pad64/16][64][8] */
  const void* pw_w16;           /* the same weights as B fragments of v_mfma_f32_16x16x32_bf16: bf16 [c_out_pad32/16][c_in_pad64/32][64][8], lane l,
                                   element j = W[16 tile + (l & 15)][32 kstep + 8 (l >> 4) + j] (plan.pack_pw_frags16).  May be NULL: the split kernel
                                   then declines and the 4 + 4-wave kernel runs on pw_w */
  const void* res_w16;          /* bf16 [c_out_pad32/16][c_res_pad64/32][64][8] */
  const float* bias;            /* f32  [c_out_pad32]  bn_shift (+ residual bn_shift) */
  const void* se_y;             /* ABI v7, may be NULL.  Squeeze-excite tail of a CitrinetBlock (citrinet/blocks.py:186-196) in THIS launch's epilogue:
                                   y = relu(se_gate[b][co] * se_y[b][co][t] + result), se_y = bf16 [B][c_out][pitch_out] (the main branch's output, tail-zero
                                   rows, 16-byte aligned), the launch being the block's residual 1x1 conv + BatchNorm (depthwise = 0, kernel = 1, stride 1,
                                   relu ignored, both TS_TCS_*TAIL* flags).  Replaces the separate ts_se_apply_fwd pass; any other configuration returns
                                   TS_EUNSUPPORTED and the caller runs the launch without se_y followed by ts_se_apply_fwd. */
  const float* se_gate;         /* f32 [B][c_out] from ts_se_gate_fwd */
  float* stats;                 /* ABI v8, may be NULL.  BatchNorm(train) statistics of the result, per TILE: f32 [c_out][batch * n_tiles][2] = (sum y, sum y^2)
                                   over the tile's frames < t_out, n_tiles = ceil(t_out / ts_tcs_pointwise_tile_frames(batch, c_out, t_out)); written by the
                                   masked pointwise-only launch (depthwise = 0, kernel 1, stride 1, bf16 result, no TS_TCS_IN_TAILZERO: what the training path's
                                   1x1 forward is) out of its accumulators; ts_train_dwconv_fwd_bn_tiles sums the tiles.  Anything else: TS_EUNSUPPORTED. */
} ts_tcs_desc;

/* ts_tcs_desc.flags
 * TS_TCS_IN_TAILZERO : x (and x_res) satisfy the tail-zero invariant -- every row is 0 from its length to the pitch,
 *   pitch >= ts_time_pitch(T) (which includes the slack the tiles over-read), and >= TS_GUARD_BYTES of zeros sit
 *   before the first and after the last row of the buffer.  Lets the kernel skip every mask and bounds check.
 * TS_TCS_OUT_ZERO_TAIL : store 0 for frames >= the output length, so that y satisfies the invariant for the next
 *   launch (the reference leaves relu(bias) there, quirk A2; use 0 only for outputs the caller never exposes).
 * TS_TCS_TAPS_PHASE : dilation-2 layers only (stride 1, even padding, both flags above set): dw_taps / dw_ksteps are packed
 *   for the phase-split form -- the even and the odd frames of a row are filtered as two dilation-1 sequences -- i.e. as
 *   for (kernel, stride 1, dilation 1, padding / 2).  TS_EUNSUPPORTED when the geometry has no phase-split kernel: launch
 *   again with the plain dilation-2 fragments. */
#define TS_TCS_IN_TAILZERO 1
#define TS_TCS_OUT_ZERO_TAIL 2
#define TS_TCS_TAPS_PHASE 4
#define TS_GUARD_BYTES 1024

/* x: bf16 [B][c_in][pitch_in]; len_in: int32 [B] valid frames of x (frames >= len are treated as 0,
 * quirk A2); x_res / len_res likewise for the residual input (may be NULL when c_res == 0);
 * y: bf16 or f32 [B][c_out][pitch_out]. */
int ts_tcs_subblock_fwd(const ts_tcs_desc* desc, const void* x, const int32_t* len_in, const void* x_res,
                        const int32_t* len_res, void* y, void* stream);


/* ------------------------------------------------------------------------------------------------
 * General f32-accumulating GEMM on the f32 matrix-core instruction (csrc/gemm_f32.hip): what the "reference arithmetic" modes run on -- the
 * f32 1x1 convolutions of the training path (quartznet/blocks.py:181 computes in f32), wav2vec2's precision = "fp32" (transformers' f32
 * linears, huggingface/compatibility.py:31-42) -- and the bf16-operand products without a kernel of their own.
 *   C[z][m][n] = sum_{j < nkb} sum_{k < K} A(z, j; m, k) B(z, j; k, n) (+ C if beta) (+ bias[n])
 *   A(m, k) at a + z sa + j ska + m a_rs + k a_cs, B(k, n) at b + z sb + j skb + k b_rs + n b_cs (ELEMENT strides): one of (a_rs, a_cs)
 *   and one of (b_rs, b_cs) is 1; C row-major, ldc >= N.  Operands with a 16-byte aligned base (8 for bf16) and strides that are multiples of 4
 *   are fetched with vector loads, others element by element.
 * in_bf16: A and B are bf16 (widened exactly), else f32; out_bf16: C is bf16, else f32.  TS_EUNSUPPORTED when neither stride of an operand is 1.
 * ---------------------------------------------------------------------------------------------- */
int ts_gemm_f32(const void* a, int64_t a_rs, int64_t a_cs, int64_t sa, int64_t ska, const void* b, int64_t b_rs, int64_t b_cs, int64_t sb,
                int64_t skb, void* c, int64_t ldc, int64_t sc, const float* bias, int32_t m, int32_t n, int32_t k, int32_t nkb, int32_t batch,
                int32_t in_bf16, int32_t out_bf16, int32_t beta, void* stream);
/* f32 operands and result, with a second batch level: grid z = z1 * batch2 + z2, operand offsets z1 s? + z2 s?2 (e.g. (clip, head)) */
int ts_gemm_f32_b2(const void* a, int64_t a_rs, int64_t a_cs, int64_t sa, int64_t sa2, int64_t ska, const void* b, int64_t b_rs, int64_t b_cs,
                   int64_t sb, int64_t sb2, int64_t skb, void* c, int64_t ldc, int64_t sc, int64_t sc2, const float* bias, int32_t m, int32_t n,
                   int32_t k, int32_t nkb, int32_t batch, int32_t batch2, int32_t beta, void* stream);

/* ------------------------------------------------------------------------------------------------
 * wav2vec2 fine-tuning (csrc/w2v_train.hip): what the backward pass of transformers.Wav2Vec2Model needs besides GEMMs (the reference trains
 * HuggingFace checkpoints through BaseCTCModule.training_step, module.py:102-127, with the conv feature extractor frozen,
 * huggingface/compatibility.py:27-28).  f32 [rows][c] activations, rows = clips x frames.
 *   ts_w2v_layernorm_bwd  y = LN(x (+ res)) gamma + beta: dx (= d res), dgamma += , dbeta += ; workspace ts_w2v_layernorm_bwd_workspace bytes
 *   ts_w2v_colsum         out[j] += sum_r x[r ld + j]   (bias gradients)
 *   ts_w2v_gelu_fwd/_bwd  y = gelu(z + bias[col]) (erf form; bias may be NULL) and dz = dy gelu'(z + bias[col])
 *   ts_w2v_softmax_fwd    s [batch][heads][t][pitch >= t] -> softmax(scale s) in place, keys >= key_len[clip] masked, columns t .. pitch zeroed
 *                         (ABI v9: a pitch that is a multiple of 4 keeps the attention products on ts_gemm_f32's vector loads); _bwd: dp -> scale p (dp - <dp, p>)
 *   ts_w2v_pad_rows       time-axis zero padding of [batch][t][c] (extract = 0) and its inverse (1)
 *   ts_w2v_mask_embed     train-time masking: rows with mask != 0 <- embed (forward); backward: dembed += masked rows of dy, which become 0
 *   ts_w2v_add            y = a + b
 * ---------------------------------------------------------------------------------------------- */
int64_t ts_w2v_layernorm_bwd_workspace(int64_t rows, int32_t c);
int ts_w2v_layernorm_bwd(const float* x, const float* res, const float* gamma, const float* dy, float eps, int64_t rows, int32_t c, float* dx,
                         float* dgamma, float* dbeta, void* workspace, void* stream);
/* The same with dgamma / dbeta WRITTEN instead of added to (the first launch zeroes them, ONE reducing launch adds the partials of both): callers that hand in fresh
 * gradient tensors need no zero fill. */
int ts_w2v_layernorm_bwd_set(const float* x, const float* res, const float* gamma, const float* dy, float eps, int64_t rows, int32_t c, float* dx,
                             float* dgamma, float* dbeta, void* workspace, void* stream);
int ts_w2v_colsum(const float* x, int64_t rows, int32_t c, int64_t ld, float* out, void* stream);
/* Operand casts of the MIXED-PRECISION fine-tuning products (ABI v11; the reference under Lightning's precision="bf16-mixed": f32 master weights,
 * f32 gradients, bf16 GEMM operands): x f32 [rows][c] (pitch ldx) -> y bf16 [rows][c] (pitch ldy) and / or yt bf16 [c][rows_pad] (pitch ldt), the
 * transposed copy with its rows index zero-padded to rows_pad (a multiple of 32: ts_gemm_nt_bf16's contraction index).  Either output may be NULL. */
int ts_w2v_cast_bf16_t(const float* x, int64_t ldx, int64_t rows, int32_t c, void* y, int64_t ldy, void* yt, int64_t ldt, int64_t rows_pad, void* stream);
/* The activation between the feed-forward linears of mixed-precision fine-tuning fused with the second linear's operand cast (transformers Wav2Vec2FeedForward:
 * intermediate_dense -> GELU -> dropout -> output_dense): y16[r][j] = bf16(dropout(gelu(z[r][j] + bias[j]))) and / or its transposed copy yt16 (as ts_w2v_cast_bf16_t:
 * rows >= `rows` zero up to rows_pad) -- the f32 activation is never stored.  Dropout: ts_train_dropout's mask over the dense [rows][c] matrix under `seed`
 * (p_drop = 0: none); c % 4 == 0.  ts_w2v_ffn_act_bwd: dz = dropout_backward(da) * gelu'(z + bias) with the mask re-drawn. */
int ts_w2v_ffn_act_cast(const float* z, const float* bias, int64_t rows, int32_t c, float p_drop, uint64_t seed, void* y_bf16, void* yt_bf16, int64_t ldt, int64_t rows_pad,
                        void* stream);
int ts_w2v_ffn_act_bwd(const float* z, const float* bias, int32_t c, const float* da, float p_drop, uint64_t seed, float* dz, int64_t n, void* stream);
/* The same launch also ADDING the column sums of x to colsum (f32 [c]; NULL: none): the bias gradient of a linear layer is the column sum of the dy being cast. */
int ts_w2v_cast_bf16_t_colsum(const float* x, int64_t ldx, int64_t rows, int32_t c, void* y, int64_t ldy, void* yt, int64_t ldt, int64_t rows_pad, float* colsum,
                              void* stream);
/* out[i] = sum_p parts[p * n + i], p in order (n % 4 == 0): the reduction behind ts_gemm_nt_bf16_splitk */
int ts_w2v_sum_parts(const float* parts, float* out, int64_t n, int32_t n_parts, void* stream);
/* the same with a bias row added: out[r][j] = sum_p parts[p][r][j] + bias[j], rows of c floats (c % 4 == 0) -- the split-K form of a linear layer's forward product */
int ts_w2v_sum_parts_bias(const float* parts, const float* bias, int32_t c, float* out, int64_t n, int32_t n_parts, void* stream);
int ts_w2v_gelu_fwd(const float* z, const float* bias, int32_t c, float* y, int64_t n, void* stream);
int ts_w2v_gelu_bwd(const float* z, const float* bias, int32_t c, const float* dy, float* dz, int64_t n, void* stream);
int ts_w2v_softmax_fwd(float* s, const int32_t* key_len, int32_t batch, int32_t heads, int32_t t, int32_t pitch, float scale, void* stream);
int ts_w2v_softmax_bwd(const float* p, float* dp, int64_t rows, int32_t t, int32_t pitch, float scale, void* stream);
int ts_w2v_pad_rows(const float* src, float* dst, int32_t batch, int32_t t, int32_t t_dst, int32_t left, int32_t c, int32_t extract, void* stream);
int ts_w2v_mask_embed(float* x, const uint8_t* mask, const float* embed, float* dembed, int64_t rows, int32_t c, void* stream);
int ts_w2v_add(const float* a, const float* b, float* y, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Squeeze-excite of the Citrinet blocks (eval): replaces SqueezeExcite.forward (citrinet/blocks.py:70-83: AdaptiveAvgPool1d
 * over ALL frames incl. padding -> Linear -> ReLU -> Linear -> sigmoid -> x * g) and the block tail
 * `out = relu(se(mconv(x)) + res(x))` (citrinet/blocks.py:186-196).
 *   y: bf16 [B][C][pitch] main-branch output of the block's last sub-block launch (no ReLU), tail zeroed
 *      (TS_TCS_OUT_ZERO_TAIL); len: int32 [B] valid frames of y; tail_y: f32 [C] the value the reference holds beyond the
 *      length (= the folded BN shift, ts_tcs_desc.bias: a masked input makes the conv output 0 there).
 *   ts_se_gate_fwd : pool_ws f32 [B][C + hidden] (workspace: the means, then the hidden activations), gate f32 [B][C];
 *                    w1 f32 [hidden][C], w2 f32 [C][hidden] (nn.Linear layout, no bias).
 *   ts_se_apply_fwd: out = act( gate * y + r ), r: bf16 [B][C][pitch_r] residual branch (pointwise launch, tail zeroed) or
 *                    NULL, tail_r its constant beyond the length; r_stride: out frame e reads r frame e * r_stride (a strided 1x1
 *                    residual conv computed at the input's frame rate: a 1x1 conv commutes with subsampling); zero_tail = 1
 *                    stores 0 for frames >= len (internal blocks), 0 stores the reference's values there (caller-visible output).
 * ---------------------------------------------------------------------------------------------- */
int ts_se_gate_fwd(const void* y, const int32_t* len, const float* tail_y, int32_t batch, int32_t channels, int32_t t,
                   int32_t pitch, int32_t hidden, const float* w1, const float* w2, float* pool_ws, float* gate,
                   void* stream);
int ts_se_apply_fwd(const void* y, const void* r, const float* gate, const int32_t* len, const float* tail_y,
                    const float* tail_r, int32_t batch, int32_t channels, int32_t t, int32_t pitch_y, int32_t pitch_r,
                    int32_t r_stride, int32_t pitch_out, int32_t relu, int32_t zero_tail, void* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Mel-filterbank front end (eval mode): pre-emphasis -> reflect-padded STFT power -> slaney mel ->
 * log -> per-(clip, mel) masked normalisation, replaces FilterbankFeatures.forward
 * (quartznet/transform.py:136-144, :186-208, :243-255, :77-92 -> blocks.py:136-149).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ts_frontend_desc {
  int32_t batch, n_samples;      /* waveform [B][n_samples] f32 */
  int32_t n_fft, hop, win_length;
  int32_t n_mels;
  float preemph;
  int32_t n_frames;              /* n_samples / hop + 1 */
  int32_t pitch_out;             /* time pitch of the feature tensor */
  const float* window;           /* f32 [n_fft] analysis window (hann centred in n_fft) */
  const float* mel_weights;      /* f32 packed non-zero filterbank weights */
  const int32_t* mel_offsets;    /* int32 [n_mels][2]: (first bin, offset into mel_weights); count = next offset - this */
  int32_t mel_nnz;
  /* training-mode extras (all zero / NULL in eval mode) */
  int32_t n_masks;               /* SpecAugment / SpecCutout rectangles applied after the normaliser (spec_augment.py:51-56, :96-101) */
  const int32_t* masks;          /* int32 [n_masks][4] = (f0, f1, t0, t1): features[b][f0:f1][t0:t1] = 0 for every clip */
  uint64_t dither_seed;          /* Philox key of the dither noise */
  float dither;                  /* > 0: DitherAudio (quartznet/transform.py:109-118): x + dither * N(0, 1) before pre-emphasis */
  int64_t* feat_len64;           /* optional (ABI v11): the frame lengths once more as int64 [B] -- what PowerSpectrum.get_sequence_length returns
                                    (quartznet/transform.py:182-184, .long()) -- so that the caller needs no launch of its own for them */
} ts_frontend_desc;

/* Stage 1: logmel f32 [B][n_frames][n_mels] (frame-major scratch) + per-(b, mel) partial sums.
 * Stage 2: normalise + mask + transpose into features bf16 [B][n_mels][pitch_out]; feat_len int32 [B].
 * wave_len: f32 or int lengths are converted by the caller to int32 samples (floor).
 * workspace: ts_frontend_workspace_bytes() bytes. */
int64_t ts_frontend_workspace_bytes(const ts_frontend_desc* desc);
int ts_mel_frontend_fwd(const ts_frontend_desc* desc, const float* wave, const int32_t* wave_len,
                        void* features, int32_t* feat_len, void* workspace, void* stream);
/* The five stage modules of the front end called ON THEIR OWN (reference quartznet/transform.py:71-255; the reference's tests drive each
 * with arbitrary parameters, tests/quartznet/test_transform_qn.py:130-260).  Reference layout, f32, generic and simple: FilterbankFeatures
 * itself never calls these -- its forward is ts_mel_frontend_fwd above.
 *   ts_fe_preemph         y[b][0] = x[b][0], y[b][n] = x[b][n] - coeff * x[b][n-1]                       ([B][n] -> [B][n])
 *   ts_fe_dither          y = x + dither * N(0, 1), the Philox stream of ts_mel_frontend_fwd's dither      ([B][n] -> [B][n])
 *   ts_fe_power_spectrum  |STFT|^2 of torch.stft(n_fft, hop, center=True, pad_mode="reflect"); window: f32 [n_fft] (the win_length
 *                         window centred in zeros); twiddle: f32 [n_fft][2] = (cos, sin)(2 pi j / n_fft); out [B][n_fft/2+1][n/hop+1]
 *   ts_fe_stft            the transform itself, out [B][n_fft/2+1][n/hop+1][2] = (re, im): convolution_stft of blocks.py:38-91 (the reference's
 *                         export-friendly replacement of torch.stft; same arguments as ts_fe_power_spectrum)
 *   ts_fe_mel             out[b][m][t] = log(sum_f fb[m][f] x[b][f][t] + 2^-24) (log_scale = 1) or the plain product; fb f32 [n_mels][n_freq]
 *   ts_fe_normalize       masked per-(clip, feature) normalisation with the padded frames in the variance (quirk A1), 0 beyond len[b] */
int ts_fe_preemph(const float* x, float* y, int32_t batch, int32_t n, float coeff, void* stream);
int ts_fe_dither(const float* x, float* y, int32_t batch, int32_t n, float dither, uint64_t seed, void* stream);
int ts_fe_power_spectrum(const float* x, const float* window, const float* twiddle, float* out, int32_t batch, int32_t n, int32_t n_fft,
                         int32_t hop, void* stream);
int ts_fe_stft(const float* x, const float* window, const float* twiddle, float* out, int32_t batch, int32_t n, int32_t n_fft, int32_t hop,
               void* stream);
int ts_fe_mel(const float* x, const float* fb, float* out, int32_t batch, int32_t n_freq, int32_t n_mels, int32_t t, int32_t log_scale,
              void* stream);
int ts_fe_normalize(const float* x, const int32_t* len, float* out, int32_t batch, int32_t features, int32_t t, float guard, void* stream);
/* Debug/parity hook: copy of the un-normalised log-mel [B][n_frames][n_mels] f32 left in workspace. */
const float* ts_frontend_logmel_ptr(const ts_frontend_desc* desc, const void* workspace);

/* ------------------------------------------------------------------------------------------------
 * Training-time augmentation.  SpecAugment / SpecCutout (quartznet/spec_augment.py:23-102): a mask is a row
 * (f0, f1, t0, t1) of an int32 device table, the same for every clip of the batch as in the reference.
 *   ts_spec_masks_draw : fills the table on the device from a Philox stream, in the reference's order -- n_cutout rectangles
 *     (frequency span, then a time span drawn with cut_FREQ_width: reference quirk, spec_augment.py:99-100), n_time time
 *     masks, n_freq frequency masks; each span is value = u * width, min = u' * (size - value), [long(min), long(min) + long(value)).
 *     (The Python mirror can instead fill the table from torch.rand(1) draws on the host, exactly as the reference does.)
 *   ts_spec_mask_apply : features[b][f0:f1][t0:t1] = 0 on a [B][channels][pitch] tensor, in place; elem_bytes = 2 (bf16, the
 *     internal layout) or 4 (f32, a reference-layout tensor handed to the standalone module).  ts_mel_frontend_fwd
 *     applies the same table inside its normaliser when ts_frontend_desc.masks is set (no extra pass).
 * ---------------------------------------------------------------------------------------------- */
int ts_spec_masks_draw(uint64_t seed, int32_t n_time, int32_t time_width, int32_t n_freq, int32_t freq_width, int32_t n_cutout,
                       int32_t cut_time_width, int32_t cut_freq_width, int32_t n_mels, int32_t n_frames, int32_t* table,
                       void* stream);
int ts_spec_mask_apply(void* features, int32_t elem_bytes, int32_t batch, int32_t channels, int32_t t, int32_t pitch,
                       const int32_t* table, int32_t n_masks, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Greedy CTC decode: argmax over classes then run-collapse (torch.unique_consecutive), replaces
 * module.py:100 + text_processing/transform.py:107-110.  logits f32 [B][V][pitch]; ids int32 [B][T]
 * (argmax per frame, lowest index wins ties); collapsed int32 [B][T] + counts int32 [B]; collapsed[b][i] = 0
 * for i >= counts[b] (the host's token-table lookup runs over whole rows).
 * ---------------------------------------------------------------------------------------------- */
int ts_greedy_decode(const float* logits, int32_t batch, int32_t n_classes, int32_t n_frames, int32_t pitch,
                     int32_t* ids, int32_t* collapsed, int32_t* counts, void* stream);

/* ------------------------------------------------------------------------------------------------
 * CTC loss forward + gradient w.r.t. the logits, replaces ctc_loss.py:36-47
 * (permute -> log_softmax -> F.ctc_loss(reduction="mean", zero_infinity=True)).
 * logits f32 [B][V][pitch]; targets int32 [B][s_max]; nll f32 [B] (per-utterance, inf -> 0);
 * grad f32 [B][V][pitch] (may be NULL: forward only); loss f32 [1].
 * workspace: ts_ctc_workspace_bytes() bytes.
 * ---------------------------------------------------------------------------------------------- */
int64_t ts_ctc_workspace_bytes(int32_t batch, int32_t n_classes, int32_t n_frames, int32_t s_max);
int ts_ctc_loss(const float* logits, int32_t batch, int32_t n_classes, int32_t n_frames, int32_t pitch,
                const int32_t* targets, int32_t s_max, const int32_t* input_len, const int32_t* target_len,
                int32_t blank, float* nll, float* loss, float* grad, void* workspace, void* stream);
/* Targets and lengths in the forms F.ctc_loss takes them (ctc_loss.py:36-47: padded [B][s_in] labels, int32 (kind 0) or int64 (1), row
 * stride targets_stride; lengths int32 / int64 / float32 / float64 (kind 0..3), truncated toward zero like the reference's `.long()`)
 * -> the int32 arrays ts_ctc_loss reads: targets_out [B][s_max] (s_max >= max(s_in, 1)) with positions >= length and ids outside
 * [0, n_classes) set to 0; an utterance with such an id below its length gets input length 0 and a target length >= 1 (infeasible:
 * loss 0 under zero_infinity, zero gradient) -- the device-side form of the ValueError host tensors get, without a host sync;
 * bad_rows_total (may be NULL): device counter incremented once per such utterance, for the caller to inspect at a point where it
 * synchronises anyway (epoch end). */
int ts_ctc_prepare(const void* targets, int32_t targets_kind, int64_t targets_stride, int32_t s_in, const void* target_len,
                   int32_t target_len_kind, const void* input_len, int32_t input_len_kind, int32_t batch, int32_t s_max,
                   int32_t n_classes, int32_t* targets_out, int32_t* target_len_out, int32_t* input_len_out, int32_t* bad_rows_total,
                   void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fine-tuning with a frozen encoder (the first phase of FinetuneCTCModule + FinetuneEncoderDecoder, finetune.py:19-88,
 * callbacks.py: the encoder stays frozen until `unfreeze_encoder_at_epoch`): backward of the 1x1 decoder
 * (blocks.py:199-216) and the optimizer step.
 *   ts_decoder_bwd: grad_logits f32 [B][V][pitch_g] (dL/dlogits, e.g. from ts_ctc_loss), x bf16 [B][C][pitch_x] (encoder
 *     output, NCT-p) -> d_weight f32 [V][C], d_bias f32 [V] (both overwritten).
 *   ts_adamw_step : torch.optim.AdamW on a flat f32 buffer of n elements (decoupled weight decay, bias correction with
 *     `step` >= 1, amsgrad off): p *= 1 - lr*wd; m,v updated; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
 * ---------------------------------------------------------------------------------------------- */
int ts_decoder_bwd(const float* grad_logits, const void* x, int32_t batch, int32_t n_classes, int32_t channels, int32_t t,
                   int32_t pitch_g, int32_t pitch_x, float* d_weight, float* d_bias, void* stream);
int ts_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int32_t step, void* stream);
/* The same update for all tensors of a parameter group in one launch.  table: device array of n_tensors entries of SIX 64-bit
 * words (param, grad, exp_avg, exp_avg_sq pointers, element count, bf16 shadow pointer or 0); max_numel = the largest element
 * count; one shared step.  A non-zero shadow receives the updated parameter rounded to bf16 (the GEMM operand copy of the
 * mixed-precision training path: no separate cast launch per layer and step). */
int ts_adamw_multi_step(const void* table, int32_t n_tensors, int64_t max_numel, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int32_t step, void* stream);
/* Wire format of the data-parallel gradient exchange (what Lightning DDP's all-reduce moves for the reference, module.py:102-127):
 * pack: wire[i] = bf16(grad[i] * scale) (scale = 1 / world, so the collective's SUM is the mean); unpack: grad[i] = f32(wire[i]).
 * Both pointers 16-byte aligned; the collective itself is RCCL's (torch.distributed). */
int ts_grad_wire_pack(const float* grad, void* wire_bf16, int64_t n, float scale, void* stream);
int ts_grad_wire_unpack(const void* wire_bf16, float* grad, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training-mode encoder ops (fine-tuning with the encoder unfrozen): one entry point per reference op and direction.
 * Activations are [B][C][pitch] rows, time contiguous, pitch a multiple of 8 elements with 16-byte (bf16) / 32-byte (f32)
 * aligned rows; columns >= T are scratch.  `act` selects the element type of every activation pointer of the call:
 * 0 = f32 (the reference's arithmetic), 1 = bf16 storage with f32 arithmetic inside the kernels (mixed precision).
 * Parameters, their gradients, statistics and lengths (int32 [B]) are f32 / int32 in both modes.
 *   depthwise MaskedConv1d (quartznet/blocks.py:169-182, groups = C): x masked by len_in; y masked by len_out when given
 *   1x1 MaskedConv1d: plain GEMMs (rocBLAS) on inputs the caller has masked with ts_train_mask_time
 *   BatchNorm1d in train mode (quartznet/blocks.py:222, statistics over all B*T frames incl. padding -- quirk A4 --,
 *     biased variance, eps) with optional fused ReLU; mean_rstd f32 [C][2] is saved for the backward
 *   residual add + ReLU (quartznet/blocks.py:332-337)
 * Workspaces: pwconv_bwd B*c_out*c_in floats; bn_fwd and bn_bwd 16*C doubles each (8 clip-group partial sums).
 * ---------------------------------------------------------------------------------------------- */
/* reference layout f32 [rows][t] (contiguous) <-> pitched activation rows: the boundary of the training path */
int ts_train_act_import(const float* src, void* dst, int64_t rows, int32_t t, int32_t pitch, int32_t act, void* stream);
int ts_train_act_export(const void* src, float* dst, int64_t rows, int32_t t, int32_t pitch, int32_t act, void* stream);
int ts_train_dwconv_fwd(const void* x, const int32_t* len_in, const int32_t* len_out, const float* w, void* y, int32_t batch,
                        int32_t channels, int32_t t_in, int32_t t_out, int32_t kernel, int32_t stride, int32_t dilation,
                        int32_t padding, int32_t pitch_in, int32_t pitch_out, int32_t act, void* stream);
/* len_out (may be NULL) in both directions: the forward zeroes y from len_out[b] on -- the re-masking the next MaskedConv1d applies
 * (quartznet/blocks.py:169-171) -- and the backward treats dy as zero there; no separate masking pass is needed then.
 * dw ACCUMULATES: dw += sum_{b,t} dy * x (float atomics over clip groups); hand in zeros for a plain gradient.  dw == NULL (here and in
 * ts_train_dwconv_bwd_bn): the weight is frozen -- only the data gradient is computed (and x is not read unless an input transform needs it).
 * dx == NULL (ts_train_dwconv_bwd, stride > 1 geometries whose two gradients are separate launches, dw given): the input needs no gradient (the stem's
 * input are the features) -- only the weight gradient is computed; TS_EINVAL for the fused stride-1 kernels, which form both in one pass. */
int ts_train_dwconv_bwd(const void* dy, const void* x, const int32_t* len_in, const int32_t* len_out, const float* w, void* dx,
                        float* dw, int32_t batch, int32_t channels, int32_t t_in, int32_t t_out, int32_t kernel, int32_t stride,
                        int32_t dilation, int32_t padding, int32_t pitch_in, int32_t pitch_out, int32_t act, void* stream);
/* Kernel choice of ts_train_dwconv_bwd / ts_train_dwconv_bwd_bn for bf16 rows in the "same" geometry with channels % 16 == 0 and odd
 * kernel <= 75: 1 (default) = both gradients on the matrix cores (v_mfma_f32_4x4x4_16b_bf16, taps rounded to bf16 like the forward's), 0 = the
 * packed-f32 FIR kernel (what f32 rows and every other geometry always take).  Process-wide; returns the previous mode. */
int ts_train_dwconv_bwd_select(int32_t mode);
/* Deterministic gradients (no reference counterpart; torch.use_deterministic_algorithms is the nearest notion): with a workspace set, the depthwise
 * backward kernels (ts_train_dwconv_bwd / _bwd_bn, every geometry with a fused data + weight gradient) store their per-workgroup sums of dw,
 * in_dbeta, in_dgamma there instead of adding them with float atomics, and a second launch adds them to the destinations in workgroup order: the
 * gradients of a step are then the same bits on every run.  workspace: f32 [n_floats] on the device, at least (workgroup rows of a launch) x
 * channels x (kernel + 2) for the largest layer (TS_EINVAL from the backward call otherwise); NULL switches the mode off.  Process-wide; the
 * workspace is reused by every launch, in stream order. */
int ts_train_set_deterministic(float* workspace, int64_t n_floats);
/* BatchNorm(train) [+ ReLU] between two repeats folded into the depthwise launches ("same" geometry only: stride 1, dilation 1, odd
 * kernel, padding (k-1)/2, even channel count; anything else TS_EUNSUPPORTED), so that the normalised tensor is never stored:
 *   ts_train_bn_stats       clip-group sums of v only (sums: 16*C doubles = [8][C][2] (sum v, sum v^2)), no apply pass
 *   ts_train_dwconv_fwd_bn  y = dwconv(mask(x)), x = relu?(gamma * (v - mean) * rstd + beta) formed while v is staged; mean / rstd come
 *                           out of in_sums inside the kernel, which also publishes in_mean_rstd (f32 [C][2], needed by the backward
 *                           calls) and applies the running-statistics update (momentum, unbiased variance, batch counter)
 *   ts_train_dwconv_bwd_bn  dw as ts_train_dwconv_bwd; g = dL/dx * (x > 0 when in_relu); in_dbeta += sum g, in_dgamma += sum g * xhat
 *                           (both ACCUMULATE: they are the parameter gradients of that BatchNorm and the sums its backward needs)
 *   ts_train_bn_bwd_sums    dv = gamma * rstd * (g - dbeta / n - xhat * dgamma / n), n = B * T */
int ts_train_bn_stats(const void* v, void* sums, int32_t batch, int32_t channels, int32_t t, int32_t pitch, int32_t act, void* stream);
int ts_train_dwconv_fwd_bn(const void* v, const void* in_sums, const float* in_gamma, const float* in_beta, float in_eps, int32_t in_relu,
                           float* in_mean_rstd, float* running_mean, float* running_var, float momentum, int64_t* num_batches_tracked,
                           const int32_t* len_in, const int32_t* len_out, const float* w, void* y, int32_t batch, int32_t channels,
                           int32_t t, int32_t kernel, int32_t padding, int32_t pitch, int32_t act, void* stream);
/* ts_train_dwconv_fwd_bn with the statistics as the per-tile pairs a pointwise launch left (ts_tcs_desc.stats): in_tile_sums f32 [C][in_tiles][2].
 * bf16 rows and >= 17 clips (the matrix-core depthwise forward, whose waves own a channel and sum its pairs cooperatively); else TS_EUNSUPPORTED. */
int ts_train_dwconv_fwd_bn_tiles(const void* v, const float* in_tile_sums, int32_t in_tiles, const float* in_gamma, const float* in_beta, float in_eps,
                                 int32_t in_relu, float* in_mean_rstd, float* running_mean, float* running_var, float momentum,
                                 int64_t* num_batches_tracked, const int32_t* len_in, const int32_t* len_out, const float* w, void* y, int32_t batch,
                                 int32_t channels, int32_t t, int32_t kernel, int32_t padding, int32_t pitch, int32_t act, void* stream);
int ts_train_dwconv_bwd_bn(const void* dy, const void* v, const float* in_mean_rstd, const float* in_gamma, const float* in_beta,
                           int32_t in_relu, const int32_t* len_in, const int32_t* len_out, const float* w, void* g, float* dw,
                           float* in_dgamma, float* in_dbeta, int32_t batch, int32_t channels, int32_t t, int32_t kernel, int32_t padding,
                           int32_t pitch, int32_t act, void* stream);
/* Block tail (quartznet/blocks.py:332-337): out = relu(BatchNorm(va) + BatchNorm(vb)), main and residual branch, from the clip-group
 * sums of both (ts_train_bn_stats) in one pass; publishes both mean_rstd, applies both running-statistics updates.  Backward: two
 * ts_train_bn_bwd calls with dy = d out, y = out, relu = 1 (the gate of the shared ReLU). */
int ts_train_bn2_add_relu_fwd(const void* va, const void* sums_a, const float* gamma_a, const float* beta_a, float eps_a, float* mean_rstd_a,
                              float* running_mean_a, float* running_var_a, float momentum_a, int64_t* num_batches_tracked_a,
                              const void* vb, const void* sums_b, const float* gamma_b, const float* beta_b, float eps_b, float* mean_rstd_b,
                              float* running_mean_b, float* running_var_b, float momentum_b, int64_t* num_batches_tracked_b,
                              void* out, int32_t batch, int32_t channels, int32_t t, int32_t pitch, int32_t act, void* stream);
/* The block tail in ONE launch each way (ABI v9): a workgroup owns a channel and keeps its batch x ceil(t / 512) row units of both branches in
 * registers between the statistics and the apply step -- no clip-group sums, no second pass.  Forward = ts_train_bn_stats x 2 +
 * ts_train_bn2_add_relu_fwd (quartznet/blocks.py:332-337 in train mode: batch statistics over all B * T frames, running-statistics update);
 * backward = ts_train_bn_bwd(relu = 1) of both branches with the shared gate taken from `out` (dgamma / dbeta are overwritten); `dout2` (may be NULL):
 * a second gradient of `out`, added to `dout` for frames < len2[clip] (len2 NULL: all frames) -- the sum ts_train_add would form when the block's
 * output feeds the next block's main and residual branch.
 * TS_EUNSUPPORTED when batch * ceil(t / 512) exceeds 32 (bf16 rows) / 16 (f32 rows): use the two-step entry points then. */
int ts_train_bn2_add_relu_chan_fwd(const void* va, const float* gamma_a, const float* beta_a, float eps_a, float* mean_rstd_a,
                                   float* running_mean_a, float* running_var_a, float momentum_a, int64_t* nbt_a, const void* vb,
                                   const float* gamma_b, const float* beta_b, float eps_b, float* mean_rstd_b, float* running_mean_b,
                                   float* running_var_b, float momentum_b, int64_t* nbt_b, void* out, int32_t batch, int32_t channels, int32_t t,
                                   int32_t pitch, int32_t act, void* stream);
int ts_train_bn2_chan_bwd(const void* dout, const void* dout2, const int32_t* len2, const void* out, const void* va, const void* vb, const float* gamma_a, const float* mean_rstd_a,
                          const float* gamma_b, const float* mean_rstd_b, void* dva, void* dvb, float* dgamma_a, float* dbeta_a, float* dgamma_b,
                          float* dbeta_b, int32_t batch, int32_t channels, int32_t t, int32_t pitch, int32_t act, void* stream);
int ts_train_bn_bwd_sums(const void* g, const void* v, const float* gamma, const float* mean_rstd, const float* dgamma, const float* dbeta,
                         void* dv, int32_t batch, int32_t channels, int32_t t, int32_t pitch, int32_t act, void* stream);
int ts_train_mask_time(const void* x, const int32_t* len, void* y, int32_t batch, int32_t channels, int32_t t, int32_t pitch_x,
                       int32_t pitch_y, int32_t act, void* stream);
/* pointwise convs.  precision: 0 = f32 operands and results; 1 = bf16 operands (u, dv; w = a bf16 copy made by ts_train_cast_bf16),
 * f32 results (the decoder's logits); 2 = bf16 operands AND bf16 results (the mixed-precision encoder).  f32 accumulation always;
 * d weight is f32 always. */
int ts_train_cast_bf16(const float* x, void* y_bf16, int64_t n, void* stream);
int ts_train_pwconv_fwd(const void* u, const void* w, void* v, int32_t batch, int32_t c_in, int32_t c_out, int32_t t, int32_t pitch_u,
                        int32_t pitch_v, int32_t precision, void* stream);
int ts_train_pwconv_bwd(const void* dv, const void* u, const void* w, void* du, float* dw, float* workspace, int32_t batch,
                        int32_t c_in, int32_t c_out, int32_t t, int32_t pitch_u, int32_t pitch_v, int32_t precision, void* stream);
/* Mixed-precision path (u, v, dv, du bf16): the forward product and the data gradient run on the inference kernel's pointwise-only
 * mode -- ts_tcs_subblock_fwd with depthwise = 0, kernel = 1, flags = TS_TCS_IN_TAILZERO, a zero bias, and pw_w = the B-fragments of W
 * (forward) or of W^T (data gradient: c_in and c_out swap roles); it wants c_in % 64 == 0 and pitches >= round_up(T, 192).
 *   ts_train_pack_pw_multi      packs, for n_tensors layers in ONE launch, W (f32 [c_out][c_in]) into both fragment sets (layout of
 *                               ts_tcs_desc.pw_w: bf16 [n_pad32/32][k_pad64/16][64][8]).  table: device array of n_tensors rows of five
 *                               64-bit words (W pointer, forward fragments pointer [n = c_out, k = c_in], backward fragments pointer
 *                               [n = c_in, k = c_out], c_out, c_in); max_groups = the largest per-layer group count
 *                               (c_out_pad32 * c_in_pad64 + c_in_pad32 * c_out_pad64) / 8.
 *   ts_train_pwconv_wgrad_mfma  dw += sum_b dv[b] . u[b]^T on hand-written MFMA kernels (csrc/train_gemm.hip): split over clip groups,
 *                               f32 partials in `workspace` (ts_train_pwconv_wgrad_workspace floats), summed onto dw by a second
 *                               launch.  len_u (may be NULL): frames >= len_u[b] of u count as zero -- the input mask of the MaskedConv1d
 *                               applied inside the product, so that the masked copy of u need not exist.  c_in, c_out multiples of 8, pitches multiples of 8 and >= round_up(T, 64), 16-byte aligned
 *                               bases; anything else returns TS_EUNSUPPORTED (callers fall back to ts_train_pwconv_bwd).
 *                               dw == NULL: the partials only ([n_parts][c_out][c_in], n_parts = workspace floats / (c_out * c_in)).
 *   ts_train_wgrad_reduce_multi dws[e][i] += sum_p parts[e][p][i], i < n[e], p < n_parts[e], for `count` layers in ceil(count / 64) launches:
 *                               a step replayed from a hipGraph parks every layer's partials and sums them once per backward piece instead
 *                               of once per layer (93 launch-bound ~8 us launches for QuartzNet15x5).  The four arrays are HOST arrays of
 *                               `count` entries, read at call time; n[e] % 4 == 0, 16-byte aligned pointers. */
int ts_train_pack_pw_multi(const void* table, int32_t n_tensors, int64_t max_groups, void* stream);
int64_t ts_train_pwconv_wgrad_workspace(int32_t batch, int32_t c_in, int32_t c_out);
int ts_train_pwconv_wgrad_mfma(const void* dv, const void* u, const int32_t* len_u, float* dw, float* workspace, int32_t batch, int32_t c_in, int32_t c_out,
                               int32_t t, int32_t pitch_u, int32_t pitch_v, void* stream);
/* One layer of ts_train_pwconv_wgrad_multi: the arguments of ts_train_pwconv_wgrad_mfma with dw = NULL (partials only, into `workspace`). */
typedef struct ts_wgrad_item {
  const void* dv; const void* u; const int32_t* len_u; float* workspace;
  int32_t batch, c_in, c_out, t, pitch_u, pitch_v;
} ts_wgrad_item;
/* The split-K partial products of `count` layers in ceil(count / 32) launches (workgroup -> (layer, tile) through prefix sums in the kernel
 * arguments): what a step replayed from hipGraphs launches once per backward piece, followed by ts_train_wgrad_reduce_multi.  `items` is a HOST
 * array read at call time. */
int ts_train_pwconv_wgrad_multi(const ts_wgrad_item* items, int32_t count, void* stream);
/* partial [c_out][c_in] tiles per layer the grouped launch writes (its workspace needs that many x c_out x c_in floats; fewer than the single-layer
 * launch, which must fill the chip with one layer's tiles): the n_parts to hand to ts_train_wgrad_reduce_multi */
int32_t ts_train_pwconv_wgrad_multi_parts(int32_t batch, int32_t c_in, int32_t c_out);
int ts_train_wgrad_reduce_multi(const void* const* parts, void* const* dws, const int64_t* n, const int32_t* n_parts, int32_t count, void* stream);
/* running_mean / running_var (both or neither, f32 [C]) and num_batches_tracked (int64 scalar, may be NULL): the module's running
 * statistics, updated in the same launch as nn.BatchNorm1d does (momentum blend, unbiased batch variance, counter + 1). */
int ts_train_bn_fwd(const void* v, const float* gamma, const float* beta, void* y, float* mean_rstd, void* workspace, int32_t batch,
                    int32_t channels, int32_t t, int32_t pitch, float eps, int32_t relu, float* running_mean, float* running_var,
                    float momentum, int64_t* num_batches_tracked, int32_t act, void* stream);
int ts_train_bn_bwd(const void* dy, const void* y, const void* v, const float* gamma, const float* mean_rstd, void* dv,
                    float* dgamma, float* dbeta, void* workspace, int32_t batch, int32_t channels, int32_t t, int32_t pitch,
                    int32_t relu, int32_t act, void* stream);
int ts_train_add_relu_fwd(const void* a, const void* b, void* out, int64_t rows, int32_t t, int32_t pitch, int32_t act, void* stream);
int ts_train_relu_bwd(const void* dout, const void* out, void* din, int64_t rows, int32_t t, int32_t pitch, int32_t act, void* stream);
/* out = a + mask(b) on activation rows: the sum of the two gradients that meet where a block input feeds both the main branch and the
 * residual branch (autograd's own accumulation would leave the row layout).  len_b (int32 [rows / channels], may be NULL): b counts
 * only up to its clip's length -- the backward of the residual MaskedConv1d's input mask, folded into the sum. */
int ts_train_add(const void* a, const void* b, const int32_t* len_b, int32_t channels, void* out, int64_t rows, int32_t t, int32_t pitch,
                 int32_t act, void* stream);
/* strided 1x1 MaskedConv1d (residual branch of a strided block: quartznet/blocks.py:301-311, citrinet/blocks.py:156-165) =
 * this mask + subsample pass followed by the pointwise GEMM.  backward = 0: y[b,c,j] = x[b,c,j*stride] if j*stride < len[b] else 0
 * (x rows of t_in, y rows of t_out); backward = 1: x is dy (t_out), y is dx (t_in), zero where the forward read nothing. */
int ts_train_subsample_mask(const void* x, const int32_t* len, void* y, int32_t batch, int32_t channels, int32_t t_in, int32_t t_out,
                            int32_t stride, int32_t backward, int32_t pitch_in, int32_t pitch_out, int32_t act, void* stream);
/* SqueezeExcite in train mode (citrinet/blocks.py:70-83), the passes over the [rows = B*C][t] activation:
 *   ts_train_se_pool   mean[row] = mean_t x[row][t]           (AdaptiveAvgPool1d(1) over ALL frames, quirk A3)
 *   ts_train_se_scale  y = x * gate[row] (+ add_mean[row] / t when add_mean != NULL: the pooled gradient, backward pass)
 *   ts_train_se_rowdot out[row] = sum_t a[row][t] * b[row][t]  (d gate = sum_t dy * x)
 * mean / gate / add_mean / out are f32 [rows].
 * The [B, C] bottleneck in between (citrinet/blocks.py:72-83: Linear(C, hidden, bias=False) -> ReLU -> Linear(hidden, C, bias=False) -> sigmoid;
 * w1 f32 [hidden][C], w2 f32 [C][hidden]) and its autograd backward (ABI v9; these replace `mean @ w1.t()`, `h @ w2.t()` and the four
 * products of their backward):
 *   ts_train_se_gate_fwd  hid[b][j] = relu(sum_i w1[j][i] mean[b][i]) (stored when hid != NULL), gate[b][c] = sigmoid(sum_j w2[c][j] hid[b][j]);
 *                         one launch, a workgroup per clip (inference keeps ts_se_gate_fwd's two wide launches: at 32 clips x 1024 channels the
 *                         per-clip form is latency-bound, 31.7 us against 12.0)
 *   ts_train_se_gate_bwd  dz = dgate * gate * (1 - gate); dhid = (hid > 0) * (dz . w2); dmean = dhid . w1; dw2 = dz^T . hid; dw1 = dhid^T . mean
 *                         (dz_ws f32 [B][C], dhid_ws f32 [B][hidden]: workspaces; dw1 / dw2 are overwritten, not accumulated)
 * TS_EUNSUPPORTED when C + hidden floats exceed the kernels' LDS budget (15 k). */
int ts_train_se_pool(const void* x, float* mean, int64_t rows, int32_t t, int32_t pitch, int32_t act, void* stream);
int ts_train_se_scale(const void* x, const float* gate, const float* add_mean, void* y, int64_t rows, int32_t t, int32_t pitch,
                      int32_t act, void* stream);
int ts_train_se_rowdot(const void* a, const void* b, float* out, int64_t rows, int32_t t, int32_t pitch, int32_t act, void* stream);
int ts_train_se_gate_fwd(const float* mean, const float* w1, const float* w2, float* hid, float* gate, int32_t batch, int32_t channels,
                         int32_t hidden, void* stream);
int ts_train_se_gate_bwd(const float* dgate, const float* gate, const float* hid, const float* mean, const float* w1, const float* w2,
                         float* dz_ws, float* dhid_ws, float* dmean, float* dw1, float* dw2, int32_t batch, int32_t channels, int32_t hidden,
                         void* stream);
/* nn.Dropout in train mode on activation rows (quartznet/blocks.py:227-228, blocks.py:238): y = x * keep / (1 - p), keep ~
 * Bernoulli(1 - p) from the Philox stream (seed, element index row * t + i) -- independent of pitch and element type; the backward
 * pass is the same call on dy with the same seed.  x and y may alias.  `nonce` (device uint64, may be NULL) is added to the seed
 * inside the kernel: a training step replayed from a hipGraph bumps it once per replay (ts_counter_add, captured in the graph),
 * so every replay draws a new mask although the by-value seed is frozen in the graph. */
int ts_train_dropout(const void* x, void* y, int64_t rows, int32_t t, int32_t pitch, float p, uint64_t seed, const uint64_t* nonce,
                     int32_t act, void* stream);
/* *counter += inc on the device (one thread); captured at the head of a graphed training step. */
int ts_counter_add(uint64_t* counter, uint64_t inc, void* stream);

/* ------------------------------------------------------------------------------------------------
 * wav2vec2 waveform normalisation, replaces Wav2Vec2Preprocess.forward (huggingface/transform.py:34-55 -> normalize_tensor,
 * blocks.py:118-153).  wave / out f32 [B][n_samples]; wave_len int32 [B] (may be NULL when mask_input = 0).
 * mask_input = 0: (x - mean) / sqrt(var_unbiased + div_guard); mask_input = 1: masked mean, sigma whose numerator runs over
 * ALL samples of the zero-masked input, (x - mean) / (sigma + div_guard), zero beyond the length.
 * workspace: ts_w2v_workspace_bytes(batch) bytes.
 * ---------------------------------------------------------------------------------------------- */
int64_t ts_w2v_workspace_bytes(int32_t batch);
int ts_w2v_preprocess(const float* wave, const int32_t* wave_len, int32_t batch, int32_t n_samples, int32_t mask_input,
                      float div_guard, float* out, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * wav2vec2 encoder (config C5), replaces the `self.original_encoder(audio, attention_mask=...)` call of
 * _HuggingFaceEncoderAdapt.forward (huggingface/compatibility.py:31-42), i.e. transformers' Wav2Vec2Model for the
 * group-norm / post-LN checkpoints (wav2vec2-base-960h, -large-960h).  One entry point per stage of that forward pass;
 * all activations f32, TIME-MAJOR [B][T][C] (the reference's final transpose(-1, -2) is a view).  GEMMs are rocBLAS calls.
 * ---------------------------------------------------------------------------------------------- */
/* `precision` (all GEMM-carrying entry points): 0 = f32 operands; 1 = bf16 operands (x / w / qkv pointers are bf16), f32
 * accumulation and f32 results.  `y_bf16` (may be NULL): a dense bf16 copy of the result for the next GEMM, written by the
 * same launch -- there is no separate cast pass. */
/* conv layer 0.  feat_extract_norm = "group": Conv1d(1, c, kernel, stride, bias=False) -> GroupNorm(c groups, eps) -> GELU.
 * gn_w == NULL ("layer" family): y = Conv1d(...) + gn_b (the conv bias, may be NULL), no normalisation, no activation.
 * wave f32 [B][n_samples]; w f32 [c][kernel]; y f32 (may be NULL if y_bf16 is given) [B][(n_samples - kernel) / stride + 1][c]. */
int64_t ts_w2v_conv0_workspace_bytes(int32_t batch, int64_t n_samples, int32_t c, int32_t kernel, int32_t stride);
int ts_w2v_conv0_fwd(const float* wave, int32_t batch, int64_t n_samples, const float* w, const float* gn_w, const float* gn_b,
                     int32_t c, int32_t kernel, int32_t stride, float eps, float* y, void* y_bf16, void* workspace, void* stream);
/* conv layers 1..6: act(Conv1d(c_in, c_out, kernel, stride) + bias); bias f32 [c_out] or NULL; act 0 = none, 1 = GELU.  x [B][t_in][c_in];
 * w_taps [c_out][kernel][c_in] (the reference's [c_out][c_in][kernel] with the last two axes swapped); y f32 [B][t_out][c_out]
 * (GEMM accumulator; when y_bf16 is given only the bf16 copy holds the result).  w_frag (may be NULL; bf16 mode): w_taps as MFMA B fragments,
 * ts_gemm_nt_pack_w(w_taps, kernel * c_in, c_out, kernel * c_in, ...) -- static weights are packed once and then bypass LDS. */
int ts_w2v_conv_fwd(const void* x, int32_t batch, int32_t t_in, int32_t c_in, const void* w_taps, const float* bias, int32_t c_out,
                    int32_t kernel, int32_t stride, int32_t act, int32_t precision, float* y, void* y_bf16, const void* w_frag, void* stream);
/* Token-major bf16 GEMM with fused epilogue (csrc/gemm_nt.hip) -- what ts_w2v_linear_fwd / ts_w2v_conv_fwd run in bf16 mode, exported for
 * tests and tools:  y[m][n] = act(sum_k x[m][k] w[n][k] + bias[n]) + res[m][n],  x: bf16 rows of pitch lda, w: bf16 [n][k] rows of pitch
 * ldw (torch.nn.Linear layout), f32 accumulation; bias / res (f32) may be NULL; gelu = 1: erf-GELU before the residual; y (f32, pitch ldc)
 * and / or y_bf16 (pitch ld16) receive the result (either may be NULL, res == y is allowed).  n % 32 == 0, k % 32 == 0, lda % 8 == 0,
 * 16-byte aligned operands; anything else returns TS_EUNSUPPORTED. */
int ts_gemm_nt_bf16(const void* x, int64_t lda, const void* w, int64_t ldw, const float* bias, const float* res, int64_t ld_res, float* y,
                    int64_t ldc, void* y_bf16, int64_t ld16, int64_t rows, int32_t n, int32_t k, int32_t gelu, void* stream);
/* The same product with the (static) weights additionally supplied as MFMA B fragments, w_frag[n / 16][k / 32][64][8] bf16 written by
 * ts_gemm_nt_pack_w (n * k elements; n % 16 == 0, k % 32 == 0): the B operand then goes from L2 straight to registers instead of through LDS. */
int ts_gemm_nt_pack_w(const void* w, int64_t ldw, int32_t n, int32_t k, void* w_frag, void* stream);
/* Split-K form (ABI v11) for products with few 256 x 256 output tiles and a long contraction -- the weight gradients of mixed-precision fine-tuning
 * (1024 x 1024 outputs over 4 000 rows = 16 tiles on 256 compute units): split z of `splits` multiplies columns [z k / splits, (z + 1) k / splits) of
 * both operands into parts[z] (f32 [rows][n], pitch n); (k / splits) % 32 == 0.  ts_w2v_sum_parts adds the parts in a fixed order. */
int ts_gemm_nt_bf16_splitk(const void* x, int64_t lda, const void* w, int64_t ldw, float* parts, int64_t rows, int32_t n, int32_t k, int32_t splits,
                           void* stream);
int ts_gemm_nt_bf16_packed(const void* x, int64_t lda, const void* w, int64_t ldw, const void* w_frag, const float* bias, const float* res,
                           int64_t ld_res, float* y, int64_t ldc, void* y_bf16, int64_t ld16, int64_t rows, int32_t n, int32_t k, int32_t gelu,
                           void* stream);
/* y[r][:n] = act(x[r][:k] W^T + bias) + res[r][:n];  W [n][k]; bias / res (f32) may be NULL; act bit 0: GELU (erf), bit 1:
 * only y_bf16 is wanted (y is then scratch space for the f32 GEMM result).  res == y (same pitch): the product is accumulated into y
 * in place (the residual stream).  lda / ldc / ld_res: row pitches in elements. */
int ts_w2v_linear_fwd(const void* x, int64_t lda, const void* w, const float* bias, const float* res, int64_t ld_res, float* y,
                      int64_t ldc, void* y_bf16, int64_t rows, int32_t n, int32_t k, int32_t act, int32_t precision, const void* w_frag, void* stream);
/* y = act(LayerNorm(x + xbias + res) * w + b) over the last dimension; x, res (may be NULL), y f32 [rows][c]; xbias f32 [c] or
 * NULL: the bias of the linear layer that produced x, applied here instead of in a pass of its own; act 0 = none, 1 = GELU.  y may be NULL when
 * y_bf16 is given (the pre-LayerNorm encoders only ever read the bf16 copy: a third of the launch's bytes less). */
int ts_w2v_layernorm_fwd(const float* x, const float* res, const float* xbias, const float* w, const float* b, float eps, int64_t rows,
                         int32_t c, int32_t act, float* y, void* y_bf16, void* stream);
/* hidden_states[~attention_mask] = 0: rows >= len[b] of x [B][t][c] become 0. */
int ts_w2v_mask_rows(float* x, int32_t batch, int32_t t, int32_t c, const int32_t* len, void* stream);
/* positional conv embedding: y = x + gelu(Conv1d(c, c, kernel, padding = kernel / 2, groups)(x) + bias), last frame of an
 * even kernel dropped (Wav2Vec2SamePadLayer).  x, y f32; w_taps [kernel][groups][c/groups (out)][c/groups (in)], weight-norm
 * already applied. */
int64_t ts_w2v_posconv_workspace_bytes(int32_t batch, int32_t t, int32_t c, int32_t kernel);
int ts_w2v_posconv_fwd(const float* x, int32_t batch, int32_t t, int32_t c, const void* w_taps, const float* bias, int32_t kernel,
                       int32_t groups, int32_t precision, float* y, void* y_bf16, void* workspace, void* stream);
/* The positional conv of mixed-precision FINE-TUNING on the same matrix-core kernel (bf16 operands, f32 accumulation; c / groups == 64):
 *   backward == 0:  y = res + gelu(conv(src) + bias), z = conv(src) (before bias and GELU: what the GELU backward needs; may be NULL); src == res == x
 *   backward == 1:  y = res + conv(src) over a copy of src padded for the TRANSPOSED conv: with w_taps = the forward's taps flipped along the kernel axis and
 *                   each [out][in] block transposed, and src = dz, res = dy, y is the data gradient dx = dy + conv^T(dz); bias NULL.
 * w_taps bf16 [kernel][groups][64][64]; src, res, y, z f32 [B][t][c]; workspace: ts_w2v_posconv_train_workspace bytes (the padded bf16 copy of src). */
int64_t ts_w2v_posconv_train_workspace(int32_t batch, int32_t t, int32_t c, int32_t kernel);
int ts_w2v_posconv_train(const float* src, const float* res, int32_t batch, int32_t t, int32_t c, const void* w_taps_bf16, const float* bias, int32_t kernel,
                         int32_t groups, int32_t backward, float* y, float* z, void* workspace, void* stream);
/* and its weight gradient: dw[j][g][o][i] = sum_b sum_t dz[b][t][64 g + o] x[b][t + j - kernel / 2][64 g + i] (frames outside [0, t) count as 0), dw f32
 * [kernel][groups][64][64] (written, not accumulated), dz and x f32 [B][t][c]; bf16 operands, f32 accumulation, one workgroup per (8 taps, group): no atomics. */
int64_t ts_w2v_posconv_wgrad_workspace(int32_t batch, int32_t t, int32_t c, int32_t kernel);
int ts_w2v_posconv_wgrad(const float* dz, const float* x, int32_t batch, int32_t t, int32_t c, int32_t kernel, int32_t groups, float* dw, void* workspace, void* stream);
/* the conv of ONE layer of Data2VecAudioPositionalConvEmbedding (transformers modeling_data2vec_audio.py, reached from
 * huggingface/compatibility.py:31-42 when the checkpoint is a data2vec-audio one -- tests/huggingface/test_module_huggingface.py:107-110):
 * y = Conv1d(c, c, kernel, padding = kernel / 2, groups)(x) + bias, last frame of an even kernel dropped; its LayerNorm (no affine) + GELU
 * are ts_w2v_layernorm_fwd(act = 1) with unit weights.  Arguments and workspace as ts_w2v_posconv_fwd. */
int ts_w2v_groupconv_fwd(const float* x, int32_t batch, int32_t t, int32_t c, const void* w_taps, const float* bias, int32_t kernel,
                         int32_t groups, int32_t precision, float* y, void* workspace, void* stream);
/* Wav2Vec2Adapter (config.add_adapter: transformers modeling_wav2vec2.py Wav2Vec2Adapter / Wav2Vec2AdapterLayer, behind the encoder in the forward
 * pass that huggingface/compatibility.py:31-42 calls): each layer is Conv1d(c, 2c, kernel, stride, padding = 1) -- ts_w2v_pad_rows + ts_w2v_conv_fwd --
 * followed by GLU over the channels: y[r][j] = x[r][j] * sigmoid(x[r][c + j]), x f32 [rows][2c], y f32 [rows][c], y_bf16 optional copy. */
int ts_w2v_glu_fwd(const float* x, int64_t rows, int32_t c, float* y, void* y_bf16, void* stream);
/* self-attention core: qkv [B][t][3c] (q | k | v, heads are contiguous column blocks), softmax(q k^T / sqrt(c / heads)) v
 * -> ctx [B][t][c]; qkv and ctx are f32 (precision 0) or bf16 (precision 1), scores and softmax f32.
 * key_len int32 [B] or NULL: keys >= key_len[b] get probability 0 (the reference's additive mask). */
int64_t ts_w2v_attention_workspace_bytes(int32_t batch, int32_t t, int32_t heads, int32_t precision);
int ts_w2v_attention_fwd(const void* qkv, int32_t batch, int32_t t, int32_t c, int32_t heads, const int32_t* key_len,
                         int32_t precision, void* ctx, void* workspace, void* stream);
/* The same core for mixed-precision FINE-TUNING (head_dim 64), fused: no [t][t] score / probability matrix exists in either direction
 * (csrc/w2v_attn_train.hip; transformers Wav2Vec2Attention under the reference's training_step, thunder module.py:102-127).
 *   ts_w2v_attention_train_fwd   ctx = dropout(softmax(q k^T / 8 + key mask)) v;  qkv bf16 [B][t][3c], ctx f32 [B][t][c], lse2 f32 [B][heads][t] = the row
 *                                statistic max + log2(sum) in the log2 domain (scale folded in) the backward rebuilds probabilities from.  Dropout: the
 *                                mask of ts_train_dropout over the logical [B * heads * t][t] probability matrix (element e = row * t + key: word e & 3 of
 *                                Philox block e >> 2 under `seed`), kept entries scaled by 1 / (1 - p_drop); drawn once per call as a bitstring into `workspace`
 *                                (ts_w2v_attention_train_fwd_workspace bytes; unused and may be NULL when p_drop = 0).
 *   ts_w2v_attention_train_bwd   dqkv f32 [B][t][3c] (every element written) from dctx f32 [B][t][c], ctx, lse2 and the same qkv / key_len / p_drop / seed;
 *                                fwd_mask: the forward call's workspace if the caller kept it, else NULL (the mask is then re-drawn from the seed);
 *                                workspace: ts_w2v_attention_train_bwd_workspace bytes (bf16 copy of dctx, the row sums D, room for the mask bits).  Two launches
 *                                that each rebuild the probabilities: no atomics, fixed summation order.
 * key_len[b] <= 0 (no valid key): every probability of the clip is 0, ctx = 0 and all three gradients 0 -- ts_w2v_softmax_fwd's convention.
 * TS_EUNSUPPORTED unless c / heads == 64. */
int64_t ts_w2v_attention_train_fwd_workspace(int32_t batch, int32_t t, int32_t c, int32_t heads);
int ts_w2v_attention_train_fwd(const void* qkv_bf16, int32_t batch, int32_t t, int32_t c, int32_t heads, const int32_t* key_len, float p_drop,
                               uint64_t seed, float* ctx, float* lse2, void* workspace, void* stream);
int64_t ts_w2v_attention_train_bwd_workspace(int32_t batch, int32_t t, int32_t c, int32_t heads);
int ts_w2v_attention_train_bwd(const void* qkv_bf16, int32_t batch, int32_t t, int32_t c, int32_t heads, const int32_t* key_len, float p_drop,
                               uint64_t seed, const float* dctx, const float* ctx, const float* lse2, const void* fwd_mask, float* dqkv, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The steps either side of the hot path, on the device.
 *   ts_audio_prep  : AudioFileLoader.preprocess_audio (data/dataset.py:49-77): mono = mean over channels, minus its mean over
 *     time, resampled by torchaudio's polyphase sinc kernel (torchaudio.functional.resample 0.12.0 defaults).  audio f32
 *     [channels][t]; kernel f32 [new][kw] or NULL (no resampling; then t_out == t); orig / new = the two rates divided by
 *     their gcd; width = the kernel's left padding; out f32 [t_out], t_out = ceil(new * t / orig);
 *     workspace: ts_audio_prep_workspace_bytes(t) bytes.
 *   ts_collate_pad : asr_collate's pad_sequence (data/dataloader_utils.py:17-33): clip_table = n_clips x {const float*, int64 len}
 *     in device memory (already sorted by the caller), out f32 [n_clips][max_len], zero padded.
 *   ts_edit_distance : Levenshtein distance of n_pairs (a, b) int32 symbol sequences given as concatenations + offsets
 *     int32 [n_pairs + 1]: what torchmetrics' CharErrorRate / WordErrorRate sum in validation_step (module.py:153-156).
 *   ts_encode_chars : BatchTextTransformer.encode for character vocabularies (text_processing/transform.py:65-92): code points
 *     int32 (concatenated rows + offsets) -> ids by binary search in vocab_cp (sorted) / vocab_id, unknown -> unk_id,
 *     start_id / end_id < 0 = none; out int64 [n_rows][s_max] padded with pad_id, lens int64 [n_rows].
 * ---------------------------------------------------------------------------------------------- */
int64_t ts_audio_prep_workspace_bytes(int64_t t);
int ts_audio_prep(const float* audio, int32_t channels, int64_t t, const float* kernel, int32_t orig, int32_t new_, int32_t kw,
                  int32_t width, float* out, int64_t t_out, void* workspace, void* stream);
int ts_collate_pad(const void* clip_table, int32_t n_clips, int64_t max_len, float* out, void* stream);
int ts_edit_distance(const int32_t* a, const int32_t* a_off, const int32_t* b, const int32_t* b_off, int32_t n_pairs,
                     int32_t max_a_len, int32_t* dist, void* stream);
int ts_encode_chars(const int32_t* text, const int32_t* off, int32_t n_rows, const int32_t* vocab_cp, const int32_t* vocab_id,
                    int32_t n_vocab, int32_t unk_id, int32_t start_id, int32_t end_id, int32_t pad_id, int32_t s_max,
                    int64_t* out, int64_t* lens, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Layout helpers at the boundary: reference-layout f32 [B][C][T] <-> NCT-p bf16 [B][C][pitch].
 * ---------------------------------------------------------------------------------------------- */
/* len (may be NULL): int32 [B]; frames >= len[b] are written as 0 so that dst satisfies the tail-zero invariant. */
/* im2col along time for the reference-valid convolutions that have no fused kernel of their own -- dense MaskedConv1d with
 * kernel_size > 1 (quartznet/blocks.py:213-221, non-separable blocks) and depthwise stride > 2 (swept by the reference's
 * tests/quartznet/test_blocks_qn.py:158-243): out bf16 [B][kernel * channels][pitch_out], row u * channels + c, frame t =
 * mask(x)[b][c][t * stride + u * dilation - padding] (0 outside [0, len[b]) -- the MaskedConv1d input mask; len may be NULL), zero from
 * t_out to the pitch.  The convolution is then ONE pointwise launch of ts_tcs_subblock_fwd over kernel * channels input channels
 * (weights [c_out][u * channels + c] = W[c_out][c][u], or pw[c_out][c] * dw[c][u] for a separable pair). */
int ts_im2col_time(const void* x, const int32_t* len, void* out, int32_t batch, int32_t channels, int32_t t_in, int32_t pitch_in,
                   int32_t kernel, int32_t stride, int32_t dilation, int32_t padding, int32_t t_out, int32_t pitch_out, void* stream);
/* Length arithmetic in one launch: out[i] = floor((in[i] + add) / div) + plus, in the arithmetic of the input type, written as `out_kind`
 * (`out` may be NULL) and, when out_i32 != NULL, once more as int32 -- what the kernels take.  kinds: 0 f32, 1 int64, 2 int32.
 * Replaces the ATen chains of MaskedConv1d.get_seq_len (quartznet/blocks.py:150-163: add = 2p - d(k-1) - 1, div = stride, plus = 1, same
 * dtype out) and PowerSpectrum.get_sequence_length (quartznet/transform.py:170-175: add = 0, div = hop, plus = 1, int64 out). */
int ts_lengths_map(const void* in, int32_t in_kind, void* out, int32_t out_kind, int32_t* out_i32, int32_t n, int64_t add, int64_t div,
                   int64_t plus, void* stream);
int ts_pack_activation(const float* src, const int32_t* len, int32_t batch, int32_t channels, int32_t t, void* dst_bf16,
                       int32_t pitch, void* stream);
int ts_unpack_activation(const void* src_bf16, int32_t batch, int32_t channels, int32_t t, int32_t pitch,
                         float* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* THUNDER_SPEECH_AMD_H */
