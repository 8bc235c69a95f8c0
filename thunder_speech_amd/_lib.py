"""ctypes binding of the C ABI declared in include/thunder_speech_amd.h.

There is deliberately NO fallback: if the HIP shared library is missing or a kernel reports an error
the caller gets a RuntimeError -- the product path never silently computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from .build import lib_path

_lib: Optional[C.CDLL] = None
ABI_VERSION = 11
TCS_IN_TAILZERO = 1
TCS_OUT_ZERO_TAIL = 2
TCS_TAPS_PHASE = 4
TS_EUNSUPPORTED = -2
GUARD_BYTES = 1024

EXPORTED_SYMBOLS = [
    "ts_abi_version", "ts_build_target", "ts_time_pitch", "ts_tcs_subblock_fwd",
    "ts_frontend_workspace_bytes", "ts_mel_frontend_fwd", "ts_frontend_logmel_ptr",
    "ts_greedy_decode", "ts_ctc_workspace_bytes", "ts_ctc_loss", "ts_ctc_prepare",
    "ts_pack_activation", "ts_unpack_activation", "ts_lengths_map", "ts_im2col_time", "ts_fe_preemph", "ts_fe_dither", "ts_fe_power_spectrum", "ts_fe_stft", "ts_fe_mel", "ts_fe_normalize", "ts_gemm_nt_bf16", "ts_gemm_nt_pack_w", "ts_gemm_nt_bf16_packed", "ts_gemm_f32", "ts_gemm_f32_b2", "ts_w2v_layernorm_bwd_workspace", "ts_w2v_layernorm_bwd", "ts_w2v_layernorm_bwd_set", "ts_w2v_colsum", "ts_w2v_cast_bf16_t", "ts_w2v_cast_bf16_t_colsum", "ts_w2v_ffn_act_cast", "ts_w2v_ffn_act_bwd", "ts_w2v_sum_parts", "ts_w2v_sum_parts_bias", "ts_gemm_nt_bf16_splitk", "ts_w2v_gelu_fwd", "ts_w2v_gelu_bwd",
    "ts_w2v_softmax_fwd", "ts_w2v_softmax_bwd", "ts_w2v_pad_rows", "ts_w2v_mask_embed", "ts_w2v_add", "ts_se_gate_fwd", "ts_se_apply_fwd",
    "ts_decoder_bwd", "ts_adamw_step", "ts_adamw_multi_step", "ts_w2v_workspace_bytes", "ts_w2v_preprocess",
    "ts_train_dwconv_fwd", "ts_train_dwconv_bwd", "ts_train_dwconv_bwd_select", "ts_train_set_deterministic", "ts_train_mask_time", "ts_train_pwconv_fwd", "ts_train_pwconv_bwd", "ts_train_pack_pw_multi", "ts_train_pwconv_wgrad_workspace", "ts_train_pwconv_wgrad_mfma", "ts_train_pwconv_wgrad_multi", "ts_train_pwconv_wgrad_multi_parts", "ts_train_wgrad_reduce_multi",
    "ts_train_cast_bf16", "ts_train_bn_fwd", "ts_train_bn_bwd", "ts_train_add_relu_fwd", "ts_train_relu_bwd",
    "ts_train_bn_stats", "ts_train_dwconv_fwd_bn", "ts_train_dwconv_fwd_bn_tiles", "ts_tcs_pointwise_tile_frames", "ts_tcs_pointwise_wide", "ts_train_dwconv_bwd_bn", "ts_train_bn_bwd_sums", "ts_train_bn2_add_relu_fwd", "ts_train_bn2_add_relu_chan_fwd", "ts_train_bn2_chan_bwd",
    "ts_w2v_conv0_workspace_bytes", "ts_w2v_conv0_fwd", "ts_w2v_conv_fwd", "ts_w2v_linear_fwd", "ts_w2v_layernorm_fwd", "ts_w2v_posconv_train_workspace", "ts_w2v_posconv_train", "ts_w2v_posconv_wgrad_workspace", "ts_w2v_posconv_wgrad", "ts_w2v_attention_train_fwd_workspace", "ts_w2v_attention_train_fwd", "ts_w2v_attention_train_bwd_workspace", "ts_w2v_attention_train_bwd",
    "ts_w2v_mask_rows", "ts_w2v_posconv_workspace_bytes", "ts_w2v_posconv_fwd", "ts_w2v_groupconv_fwd", "ts_w2v_glu_fwd", "ts_w2v_attention_workspace_bytes",
    "ts_w2v_attention_fwd",
    "ts_spec_masks_draw", "ts_spec_mask_apply", "ts_train_dropout", "ts_counter_add", "ts_train_add", "ts_train_act_import", "ts_train_act_export",
    "ts_audio_prep_workspace_bytes", "ts_audio_prep", "ts_collate_pad", "ts_edit_distance", "ts_encode_chars",
    "ts_train_subsample_mask", "ts_train_se_pool", "ts_train_se_scale", "ts_train_se_rowdot", "ts_train_se_gate_fwd", "ts_train_se_gate_bwd",
    "ts_grad_wire_pack", "ts_grad_wire_unpack",
]


class TcsDesc(C.Structure):
    """struct ts_tcs_desc"""
    _fields_ = [
        ("batch", C.c_int32), ("c_in", C.c_int32), ("c_out", C.c_int32), ("t_in", C.c_int32), ("t_out", C.c_int32),
        ("pitch_in", C.c_int32), ("pitch_out", C.c_int32),
        ("kernel", C.c_int32), ("stride", C.c_int32), ("dilation", C.c_int32), ("padding", C.c_int32),
        ("depthwise", C.c_int32), ("relu", C.c_int32), ("out_fp32", C.c_int32),
        ("c_res", C.c_int32), ("pitch_res", C.c_int32), ("t_res", C.c_int32), ("res_stride", C.c_int32),
        ("dw_ksteps", C.c_int32), ("flags", C.c_int32),
        ("dw_taps", C.c_void_p), ("dw_taps_raw", C.c_void_p), ("pw_w", C.c_void_p), ("res_w", C.c_void_p), ("pw_w16", C.c_void_p), ("res_w16", C.c_void_p), ("bias", C.c_void_p),
        ("se_y", C.c_void_p), ("se_gate", C.c_void_p), ("stats", C.c_void_p),
    ]


class WgradItem(C.Structure):
    """struct ts_wgrad_item"""
    _fields_ = [("dv", C.c_void_p), ("u", C.c_void_p), ("len_u", C.c_void_p), ("workspace", C.c_void_p),
                ("batch", C.c_int32), ("c_in", C.c_int32), ("c_out", C.c_int32), ("t", C.c_int32), ("pitch_u", C.c_int32), ("pitch_v", C.c_int32)]


class FrontendDesc(C.Structure):
    """struct ts_frontend_desc"""
    _fields_ = [
        ("batch", C.c_int32), ("n_samples", C.c_int32), ("n_fft", C.c_int32), ("hop", C.c_int32),
        ("win_length", C.c_int32), ("n_mels", C.c_int32), ("preemph", C.c_float), ("n_frames", C.c_int32),
        ("pitch_out", C.c_int32),
        ("window", C.c_void_p), ("mel_weights", C.c_void_p), ("mel_offsets", C.c_void_p), ("mel_nnz", C.c_int32),
        ("n_masks", C.c_int32), ("masks", C.c_void_p), ("dither_seed", C.c_uint64), ("dither", C.c_float), ("feat_len64", C.c_void_p),
    ]


def is_available() -> bool:
    return os.path.exists(lib_path())


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"thunder_speech_amd: HIP extension {path} is missing. Build it with "
            "`python -m thunder_speech_amd.build` (needs hipcc / ROCm); there is no CPU fallback.")
    import torch  # noqa: F401  -- must initialise its bundled HIP runtime BEFORE our code object is loaded
    L = C.CDLL(path)
    missing = [s for s in EXPORTED_SYMBOLS if not hasattr(L, s)]
    if missing:
        raise RuntimeError(f"thunder_speech_amd: {path} does not export {missing}; rebuild it")
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.ts_abi_version.restype = C.c_int
    L.ts_build_target.restype = C.c_char_p
    L.ts_time_pitch.argtypes = [C.c_int]
    L.ts_time_pitch.restype = C.c_int
    L.ts_tcs_subblock_fwd.argtypes = [C.POINTER(TcsDesc), vp, vp, vp, vp, vp, vp]
    L.ts_tcs_subblock_fwd.restype = C.c_int
    L.ts_frontend_workspace_bytes.argtypes = [C.POINTER(FrontendDesc)]
    L.ts_frontend_workspace_bytes.restype = i64
    L.ts_mel_frontend_fwd.argtypes = [C.POINTER(FrontendDesc), vp, vp, vp, vp, vp, vp]
    L.ts_mel_frontend_fwd.restype = C.c_int
    L.ts_frontend_logmel_ptr.argtypes = [C.POINTER(FrontendDesc), vp]
    L.ts_frontend_logmel_ptr.restype = vp
    L.ts_greedy_decode.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, vp]
    L.ts_greedy_decode.restype = C.c_int
    L.ts_ctc_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.ts_ctc_workspace_bytes.restype = i64
    L.ts_ctc_loss.argtypes = [vp, i32, i32, i32, i32, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp]
    L.ts_ctc_loss.restype = C.c_int
    L.ts_ctc_prepare.argtypes = [vp, i32, i64, i32, vp, i32, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    L.ts_ctc_prepare.restype = C.c_int
    L.ts_pack_activation.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp]
    L.ts_pack_activation.restype = C.c_int
    L.ts_unpack_activation.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    L.ts_unpack_activation.restype = C.c_int
    L.ts_se_gate_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    L.ts_se_gate_fwd.restype = C.c_int
    L.ts_se_apply_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    L.ts_se_apply_fwd.restype = C.c_int
    L.ts_decoder_bwd.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    L.ts_decoder_bwd.restype = C.c_int
    f32 = C.c_float
    L.ts_adamw_step.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp]
    L.ts_adamw_step.restype = C.c_int
    L.ts_adamw_multi_step.argtypes = [vp, i32, i64, f32, f32, f32, f32, f32, i32, vp]
    L.ts_adamw_multi_step.restype = C.c_int
    L.ts_w2v_workspace_bytes.argtypes = [i32]
    L.ts_w2v_workspace_bytes.restype = i64
    L.ts_w2v_preprocess.argtypes = [vp, vp, i32, i32, i32, f32, vp, vp, vp]
    L.ts_w2v_preprocess.restype = C.c_int
    L.ts_w2v_conv0_workspace_bytes.argtypes = [i32, i64, i32, i32, i32]
    L.ts_w2v_conv0_workspace_bytes.restype = i64
    L.ts_w2v_conv0_fwd.argtypes = [vp, i32, i64, vp, vp, vp, i32, i32, i32, f32, vp, vp, vp, vp]
    L.ts_w2v_conv_fwd.argtypes = [vp, i32, i32, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    L.ts_w2v_linear_fwd.argtypes = [vp, i64, vp, vp, vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp, vp]
    L.ts_w2v_layernorm_fwd.argtypes = [vp, vp, vp, vp, vp, f32, i64, i32, i32, vp, vp, vp]
    L.ts_w2v_attention_train_fwd.argtypes = [vp, i32, i32, i32, i32, vp, f32, C.c_uint64, vp, vp, vp, vp]
    L.ts_w2v_attention_train_fwd_workspace.argtypes = [i32, i32, i32, i32]
    L.ts_w2v_attention_train_fwd_workspace.restype = i64
    L.ts_w2v_posconv_train_workspace.argtypes = [i32, i32, i32, i32]
    L.ts_w2v_posconv_train_workspace.restype = i64
    L.ts_w2v_posconv_train.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.ts_w2v_posconv_wgrad_workspace.argtypes = [i32, i32, i32, i32]
    L.ts_w2v_posconv_wgrad_workspace.restype = i64
    L.ts_w2v_posconv_wgrad.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
    L.ts_w2v_attention_train_bwd_workspace.argtypes = [i32, i32, i32, i32]
    L.ts_w2v_attention_train_bwd_workspace.restype = i64
    L.ts_w2v_attention_train_bwd.argtypes = [vp, i32, i32, i32, i32, vp, f32, C.c_uint64, vp, vp, vp, vp, vp, vp, vp]
    L.ts_w2v_mask_rows.argtypes = [vp, i32, i32, i32, vp, vp]
    L.ts_w2v_posconv_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.ts_w2v_posconv_workspace_bytes.restype = i64
    L.ts_w2v_posconv_fwd.argtypes = [vp, i32, i32, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.ts_w2v_groupconv_fwd.argtypes = [vp, i32, i32, i32, vp, vp, i32, i32, i32, vp, vp, vp]
    L.ts_w2v_glu_fwd.argtypes = [vp, i64, i32, vp, vp, vp]
    L.ts_w2v_attention_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.ts_w2v_attention_workspace_bytes.restype = i64
    L.ts_w2v_attention_fwd.argtypes = [vp, i32, i32, i32, i32, vp, i32, vp, vp, vp]
    for fn in ("ts_w2v_conv0_fwd", "ts_w2v_conv_fwd", "ts_w2v_linear_fwd", "ts_w2v_layernorm_fwd", "ts_w2v_mask_rows",
               "ts_w2v_posconv_fwd", "ts_w2v_groupconv_fwd", "ts_w2v_glu_fwd", "ts_w2v_attention_fwd"):
        getattr(L, fn).restype = C.c_int
    L.ts_train_act_import.argtypes = [vp, vp, i64, i32, i32, i32, vp]
    L.ts_train_act_export.argtypes = [vp, vp, i64, i32, i32, i32, vp]
    L.ts_train_dwconv_fwd.argtypes = [vp, vp, vp, vp, vp] + [i32] * 11 + [vp]
    L.ts_train_dwconv_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp] + [i32] * 11 + [vp]
    L.ts_train_dwconv_bwd_select.argtypes = [i32]
    L.ts_train_dwconv_bwd_select.restype = C.c_int
    L.ts_train_set_deterministic.argtypes = [vp, i64]
    L.ts_train_set_deterministic.restype = C.c_int
    L.ts_train_mask_time.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.ts_train_pwconv_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    L.ts_train_cast_bf16.argtypes = [vp, vp, i64, vp]
    L.ts_train_pwconv_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    L.ts_train_bn_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, vp, f32, vp, i32, vp]
    L.ts_train_bn_stats.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    L.ts_train_dwconv_fwd_bn.argtypes = [vp, vp, vp, vp, f32, i32, vp, vp, vp, f32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    L.ts_train_dwconv_fwd_bn_tiles.argtypes = [vp, vp, i32, vp, vp, f32, i32, vp, vp, vp, f32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    L.ts_train_dwconv_fwd_bn_tiles.restype = C.c_int
    L.ts_tcs_pointwise_tile_frames.argtypes = [i32, i32, i32]
    L.ts_tcs_pointwise_tile_frames.restype = C.c_int
    L.ts_tcs_pointwise_wide.argtypes = [i32]
    L.ts_tcs_pointwise_wide.restype = C.c_int
    L.ts_train_dwconv_bwd_bn.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    L.ts_train_bn_bwd_sums.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.ts_train_bn2_add_relu_fwd.argtypes = [vp, vp, vp, vp, f32, vp, vp, vp, f32, vp] * 2 + [vp, i32, i32, i32, i32, i32, vp]
    L.ts_train_bn2_add_relu_chan_fwd.argtypes = [vp, vp, vp, f32, vp, vp, vp, f32, vp] * 2 + [vp, i32, i32, i32, i32, i32, vp]
    L.ts_train_bn2_chan_bwd.argtypes = [vp] * 16 + [i32, i32, i32, i32, i32, vp]
    L.ts_train_bn2_add_relu_chan_fwd.restype = L.ts_train_bn2_chan_bwd.restype = C.c_int
    L.ts_train_pack_pw_multi.argtypes = [vp, i32, i64, vp]
    L.ts_train_pwconv_wgrad_workspace.argtypes = [i32, i32, i32]
    L.ts_train_pwconv_wgrad_workspace.restype = C.c_int64
    L.ts_train_pwconv_wgrad_mfma.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.ts_train_wgrad_reduce_multi.argtypes = [vp, vp, vp, vp, i32, vp]
    L.ts_train_pwconv_wgrad_multi.argtypes = [vp, i32, vp]
    L.ts_train_pwconv_wgrad_multi_parts.argtypes = [i32, i32, i32]
    L.ts_train_pwconv_wgrad_multi_parts.restype = C.c_int32
    L.ts_train_pwconv_wgrad_multi.restype = C.c_int
    L.ts_train_wgrad_reduce_multi.restype = C.c_int
    L.ts_train_bn_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.ts_train_add_relu_fwd.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp]
    L.ts_train_relu_bwd.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp]
    u64 = C.c_uint64
    L.ts_train_subsample_mask.argtypes = [vp, vp, vp] + [i32] * 9 + [vp]
    L.ts_train_se_pool.argtypes = [vp, vp, i64, i32, i32, i32, vp]
    L.ts_train_se_scale.argtypes = [vp, vp, vp, vp, i64, i32, i32, i32, vp]
    L.ts_train_se_rowdot.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp]
    L.ts_train_se_gate_fwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.ts_train_se_gate_bwd.argtypes = [vp] * 11 + [i32, i32, i32, vp]
    L.ts_train_dropout.argtypes = [vp, vp, i64, i32, i32, f32, u64, vp, i32, vp]
    L.ts_counter_add.argtypes = [vp, u64, vp]
    L.ts_lengths_map.argtypes = [vp, i32, vp, i32, vp, i32, i64, i64, i64, vp]
    L.ts_lengths_map.restype = C.c_int
    L.ts_im2col_time.argtypes = [vp, vp, vp] + [i32] * 10 + [vp]
    L.ts_im2col_time.restype = C.c_int
    L.ts_fe_preemph.argtypes = [vp, vp, i32, i32, C.c_float, vp]
    L.ts_fe_dither.argtypes = [vp, vp, i32, i32, C.c_float, C.c_uint64, vp]
    L.ts_fe_power_spectrum.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.ts_fe_stft.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.ts_fe_mel.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.ts_fe_normalize.argtypes = [vp, vp, vp, i32, i32, i32, C.c_float, vp]
    L.ts_gemm_nt_bf16.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, i64, vp, i64, i64, i32, i32, i32, vp]
    L.ts_gemm_nt_bf16.restype = C.c_int
    L.ts_gemm_f32.argtypes = [vp, i64, i64, i64, i64, vp, i64, i64, i64, i64, vp, i64, i64, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    L.ts_gemm_f32.restype = C.c_int
    L.ts_gemm_f32_b2.argtypes = [vp, i64, i64, i64, i64, i64, vp, i64, i64, i64, i64, i64, vp, i64, i64, i64, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    L.ts_w2v_layernorm_bwd_workspace.argtypes = [i64, i32]
    L.ts_w2v_layernorm_bwd_workspace.restype = i64
    L.ts_w2v_layernorm_bwd.argtypes = [vp, vp, vp, vp, C.c_float, i64, i32, vp, vp, vp, vp, vp]
    L.ts_w2v_layernorm_bwd_set.argtypes = [vp, vp, vp, vp, C.c_float, i64, i32, vp, vp, vp, vp, vp]
    L.ts_w2v_colsum.argtypes = [vp, i64, i32, i64, vp, vp]
    L.ts_w2v_cast_bf16_t.argtypes = [vp, i64, i64, i32, vp, i64, vp, i64, i64, vp]
    L.ts_w2v_cast_bf16_t_colsum.argtypes = [vp, i64, i64, i32, vp, i64, vp, i64, i64, vp, vp]
    L.ts_w2v_ffn_act_cast.argtypes = [vp, vp, i64, i32, f32, C.c_uint64, vp, vp, i64, i64, vp]
    L.ts_w2v_ffn_act_bwd.argtypes = [vp, vp, i32, vp, f32, C.c_uint64, vp, i64, vp]
    L.ts_w2v_cast_bf16_t.restype = C.c_int
    L.ts_w2v_sum_parts.argtypes = [vp, vp, i64, i32, vp]
    L.ts_w2v_sum_parts_bias.argtypes = [vp, vp, i32, vp, i64, i32, vp]
    L.ts_w2v_sum_parts.restype = C.c_int
    L.ts_gemm_nt_bf16_splitk.argtypes = [vp, i64, vp, i64, vp, i64, i32, i32, i32, vp]
    L.ts_gemm_nt_bf16_splitk.restype = C.c_int
    L.ts_w2v_gelu_fwd.argtypes = [vp, vp, i32, vp, i64, vp]
    L.ts_w2v_gelu_bwd.argtypes = [vp, vp, i32, vp, vp, i64, vp]
    L.ts_w2v_softmax_fwd.argtypes = [vp, vp, i32, i32, i32, i32, C.c_float, vp]
    L.ts_w2v_softmax_bwd.argtypes = [vp, vp, i64, i32, i32, C.c_float, vp]
    L.ts_w2v_pad_rows.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.ts_w2v_mask_embed.argtypes = [vp, vp, vp, vp, i64, i32, vp]
    L.ts_w2v_add.argtypes = [vp, vp, vp, i64, vp]
    for fn in ("ts_gemm_f32_b2", "ts_w2v_layernorm_bwd", "ts_w2v_colsum", "ts_w2v_cast_bf16_t", "ts_w2v_sum_parts", "ts_gemm_nt_bf16_splitk", "ts_w2v_gelu_fwd", "ts_w2v_gelu_bwd", "ts_w2v_softmax_fwd", "ts_w2v_softmax_bwd",
               "ts_w2v_pad_rows", "ts_w2v_mask_embed", "ts_w2v_add"):
        getattr(L, fn).restype = C.c_int
    L.ts_gemm_nt_pack_w.argtypes = [vp, i64, i32, i32, vp, vp]
    L.ts_gemm_nt_pack_w.restype = C.c_int
    L.ts_gemm_nt_bf16_packed.argtypes = [vp, i64, vp, i64, vp, vp, vp, i64, vp, i64, vp, i64, i64, i32, i32, i32, vp]
    L.ts_gemm_nt_bf16_packed.restype = C.c_int
    for f in (L.ts_fe_preemph, L.ts_fe_dither, L.ts_fe_power_spectrum, L.ts_fe_stft, L.ts_fe_mel, L.ts_fe_normalize):
        f.restype = C.c_int
    L.ts_train_add.argtypes = [vp, vp, vp, i32, vp, i64, i32, i32, i32, vp]
    for fn in ("ts_train_act_import", "ts_train_act_export", "ts_train_dwconv_fwd", "ts_train_dwconv_bwd", "ts_train_mask_time",
               "ts_train_pwconv_fwd", "ts_train_pwconv_bwd", "ts_train_pack_pw_multi", "ts_train_pwconv_wgrad_mfma", "ts_train_bn_stats", "ts_train_dwconv_fwd_bn", "ts_train_dwconv_bwd_bn", "ts_train_bn_bwd_sums", "ts_train_bn2_add_relu_fwd", "ts_train_cast_bf16", "ts_train_bn_fwd", "ts_train_bn_bwd",
               "ts_train_add_relu_fwd", "ts_train_relu_bwd", "ts_train_subsample_mask", "ts_train_se_pool", "ts_train_se_scale",
               "ts_train_se_rowdot", "ts_train_se_gate_fwd", "ts_train_se_gate_bwd", "ts_train_dropout", "ts_counter_add", "ts_train_add"):
        getattr(L, fn).restype = C.c_int
    L.ts_spec_masks_draw.argtypes = [u64] + [i32] * 9 + [vp, vp]
    L.ts_spec_mask_apply.argtypes = [vp, i32, i32, i32, i32, i32, vp, i32, vp]
    L.ts_audio_prep_workspace_bytes.argtypes = [i64]
    L.ts_audio_prep_workspace_bytes.restype = i64
    L.ts_audio_prep.argtypes = [vp, i32, i64, vp, i32, i32, i32, i32, vp, i64, vp, vp]
    L.ts_collate_pad.argtypes = [vp, i32, i64, vp, vp]
    L.ts_edit_distance.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp]
    L.ts_encode_chars.argtypes = [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    for fn in ("ts_spec_masks_draw", "ts_spec_mask_apply", "ts_audio_prep", "ts_collate_pad", "ts_edit_distance", "ts_encode_chars"):
        getattr(L, fn).restype = C.c_int
    L.ts_grad_wire_pack.argtypes = [vp, vp, i64, f32, vp]
    L.ts_grad_wire_unpack.argtypes = [vp, vp, i64, vp]
    L.ts_grad_wire_pack.restype = L.ts_grad_wire_unpack.restype = C.c_int
    if L.ts_abi_version() != ABI_VERSION:
        raise RuntimeError("thunder_speech_amd: ABI version mismatch between the Python binding and the .so")
    _lib = L
    return L


CALLS = 0          # C-ABI calls checked so far (bench.py reports calls per training step)


def check(status: int, what: str) -> None:
    global CALLS
    CALLS += 1
    if status == 0:
        return
    if status == -1:
        raise RuntimeError(f"{what}: invalid argument (TS_EINVAL)")
    if status == -2:
        raise NotImplementedError(f"{what}: configuration not supported by the HIP kernels (TS_EUNSUPPORTED)")
    raise RuntimeError(f"{what}: HIP error {status}")


def time_pitch(t: int) -> int:
    """Python mirror of ts_time_pitch (kept in sync by tests/test_capi.py)."""
    return ((max(int(t), 1) + 384 + 127) // 128) * 128
