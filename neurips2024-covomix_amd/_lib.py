"""ctypes binding of libcovomix_hip.so (the C ABI declared in include/covomix_hip.h).

There is NO fallback: if the shared library has not been built (or cannot be loaded)
every op raises.  `torch` must be imported first so that the library's
`libamdhip64.so.7` dependency resolves to the HIP runtime PyTorch already loaded
(one runtime per process => torch's streams are valid stream handles here).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime the kernels share with torch)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CVX_LIB_PATH") or os.path.join(_HERE, "libcovomix_hip.so")      # CVX_LIB_PATH: dev A/B builds

_f32p = C.POINTER(C.c_float)
ABI_VERSION = 107          # == cvx_version(): bumped whenever an argument struct or an entry point's meaning changes


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("A2", C.c_void_p), ("lda2", C.c_int64), ("K1", C.c_int32),
        ("W", C.c_void_p), ("ldw", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("act", C.c_int32),
        ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p), ("rope_T", C.c_int32), ("rope_cols", C.c_int32),
    ]


class GemmSplitIO(C.Structure):
    _fields_ = [
        ("A_hi", C.c_void_p), ("A_lo", C.c_void_p), ("lda_h", C.c_int64),
        ("A2_hi", C.c_void_p), ("A2_lo", C.c_void_p), ("lda2_h", C.c_int64),
        ("C_hi", C.c_void_p), ("C_lo", C.c_void_p), ("ldc_h", C.c_int64),
        ("write_f32", C.c_int32),
        ("Vt_hi", C.c_void_p), ("Vt_lo", C.c_void_p), ("vt_ld", C.c_int64),
        ("workspace", C.c_void_p), ("workspace_floats", C.c_int64),
        ("w_interleaved", C.c_int32),
        ("flags", C.c_int32),
        ("a_scale_dev", C.c_void_p), ("c_scale_dev", C.c_void_p), ("vt_scale_dev", C.c_void_p),
        ("c_gamma_dev", C.c_void_p), ("c_rowsq", C.c_void_p), ("c_rowsq_ld", C.c_int64), ("a_row_scale_dev", C.c_void_p),
        ("R_hi", C.c_void_p), ("R_lo", C.c_void_p), ("ldr_h", C.c_int64), ("r_scale_dev", C.c_void_p),
        ("a2_scale_dev", C.c_void_p),
    ]


class GemmNorm(C.Structure):
    _fields_ = [
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("Y_hi", C.c_void_p), ("Y_lo", C.c_void_p), ("ldy_h", C.c_int64),
        ("y_scale_dev", C.c_void_p),
        ("scale", C.c_float), ("eps", C.c_float),
    ]


class ItemLengths(C.Structure):
    _fields_ = [("item_len_dev", C.c_void_p), ("mul", C.c_int32), ("add", C.c_int32)]


class ConvArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("B", C.c_int32), ("Cin", C.c_int32), ("Lin", C.c_int32),
        ("Wp", C.c_void_p), ("bias", C.c_void_p),
        ("out", C.c_void_p), ("Cout", C.c_int32), ("Lout", C.c_int32),
        ("ksize", C.c_int32), ("dil", C.c_int32), ("pad", C.c_int32), ("up", C.c_int32),
        ("in_slope", C.c_float),
        ("res", C.c_void_p), ("accum", C.c_void_p), ("out_scale", C.c_float),
        ("items", ItemLengths),
    ]


class Conv16Args(C.Structure):
    _fields_ = [
        ("z_hi", C.c_void_p), ("z_lo", C.c_void_p),
        ("B", C.c_int32), ("L", C.c_int32), ("Lp", C.c_int32), ("Cp_in", C.c_int32), ("halo_l", C.c_int32),
        ("w_hi", C.c_void_p), ("w_lo", C.c_void_p), ("acc_scale", C.c_float), ("bias", C.c_void_p),
        ("Np", C.c_int32), ("ksize", C.c_int32), ("dil", C.c_int32),
        ("res", C.c_void_p), ("accum", C.c_void_p), ("out_x", C.c_void_p), ("out_scale", C.c_float),
        ("out_zhi", C.c_void_p), ("out_zlo", C.c_void_p), ("z_slope", C.c_float),
        ("z_scale_dev", C.c_void_p),
        ("items", ItemLengths),
    ]


class ConvT16Args(C.Structure):
    _fields_ = [
        ("z_hi", C.c_void_p), ("z_lo", C.c_void_p),
        ("B", C.c_int32), ("L_in", C.c_int32), ("Lp_in", C.c_int32), ("Cp_in", C.c_int32), ("halo_in", C.c_int32),
        ("w_hi", C.c_void_p), ("w_lo", C.c_void_p), ("acc_scale", C.c_float), ("bias", C.c_void_p),
        ("Np_out", C.c_int32), ("stride", C.c_int32), ("n_tiles", C.c_int32), ("tile_np", C.c_int32),
        ("tile_taps", C.c_int32 * 8), ("tile_pad", C.c_int32 * 8), ("tile_w_off", C.c_int64 * 8),
        ("out", C.c_void_p), ("L_out", C.c_int32), ("Lp_out", C.c_int32), ("halo_out", C.c_int32),
        ("z_scale_dev", C.c_void_p), ("amax_bits_dev", C.c_void_p),
        ("items", ItemLengths),
    ]


class Conv16Weights(C.Structure):
    _fields_ = [("w_hi", C.c_void_p), ("w_lo", C.c_void_p), ("acc_scale", C.c_float), ("bias", C.c_void_p)]


class Resblock16Args(C.Structure):
    _fields_ = [("x", C.c_void_p), ("z_hi", C.c_void_p), ("z_lo", C.c_void_p),
                ("B", C.c_int32), ("L", C.c_int32), ("Lp", C.c_int32), ("Np", C.c_int32), ("halo_l", C.c_int32),
                ("c1", Conv16Weights * 3), ("c2", Conv16Weights * 3),
                ("ksize", C.c_int32), ("dil", C.c_int32 * 3),
                ("t_hi", C.c_void_p), ("t_lo", C.c_void_p),
                ("xa", C.c_void_p), ("za_hi", C.c_void_p), ("za_lo", C.c_void_p),
                ("xb", C.c_void_p), ("zb_hi", C.c_void_p), ("zb_lo", C.c_void_p),
                ("accum", C.c_void_p), ("out", C.c_void_p), ("out_scale", C.c_float),
                ("z_scale_dev", C.c_void_p), ("items", ItemLengths)]


class Respair16Args(C.Structure):
    _fields_ = [("x", C.c_void_p),
                ("B", C.c_int32), ("L", C.c_int32), ("Lp", C.c_int32), ("Np", C.c_int32), ("halo_l", C.c_int32),
                ("c1", Conv16Weights), ("c2", Conv16Weights),
                ("ksize", C.c_int32), ("dil", C.c_int32),
                ("accum", C.c_void_p), ("out", C.c_void_p), ("out_scale", C.c_float),
                ("z_scale_dev", C.c_void_p), ("flags", C.c_int32), ("items", ItemLengths)]


class Ctx(C.Structure):
    """cvx_ctx: the launch context every entry point takes (stream, caller-owned saturation flag, CUs of the stream, flags)"""
    _fields_ = [("stream", C.c_void_p), ("sat_flag", C.c_void_p), ("n_cus", C.c_int32), ("flags", C.c_int32)]


CTX_NO_SATURATION_FLAG = 1


class T2SLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("gamma_s", "wqkv_s", "wo_s", "gamma_c", "wq_c", "wo_c", "kv_c",
                                          "gamma_f", "w1", "b1", "w2", "b2", "k_cache", "v_cache")]


class T2SDecoder(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "inner", "heads", "ff_inner", "ff_inner_pad", "depth", "streams", "vocab",
                                         "dim_emb", "n_ctx", "max_len", "top_k", "batch", "ctx_rows")] + \
               [("temperature", C.c_float), ("layers", C.POINTER(T2SLayer))] + \
               [(n, C.c_void_p) for n in ("final_gamma", "emb", "rope_cos", "rope_sin", "uniforms",
                                          "x", "q", "att", "h", "logits", "tokens", "state")] + [("cfg_scale", C.c_float)] + \
               [("uniform_steps", C.c_int32), ("queue", C.c_void_p), ("dialogues", C.c_void_p), ("start", C.c_void_p),
                ("group_loop", C.c_int32), ("pairs_per_wave", C.c_int32)]


class ResblockArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("B", C.c_int32), ("C", C.c_int32), ("L", C.c_int32),
                ("Wp1", C.c_void_p * 3), ("b1", C.c_void_p * 3), ("Wp2", C.c_void_p * 3), ("b2", C.c_void_p * 3),
                ("ksize", C.c_int32), ("dil", C.c_int32 * 3),
                ("tmp", C.c_void_p), ("out", C.c_void_p), ("accum", C.c_void_p), ("out_scale", C.c_float)]


class Linear(C.Structure):
    _fields_ = [("w", C.c_void_p), ("w_hi", C.c_void_p), ("w_lo", C.c_void_p), ("inv_scale", C.c_float),
                ("bias", C.c_void_p), ("N", C.c_int32), ("K", C.c_int32)]


class HubertLayer(C.Structure):
    _fields_ = [("qkv", Linear), ("out", Linear), ("fc1", Linear), ("fc2", Linear)] + \
               [(n, C.c_void_p) for n in ("ln1_g", "ln1_b", "ln2_g", "ln2_b")]


class HubertModel(C.Structure):
    _fields_ = [("n_conv", C.c_int32), ("conv_k", C.c_int32 * 8), ("conv_stride", C.c_int32 * 8), ("conv_c", C.c_int32 * 8),
                ("conv0_w", C.c_void_p), ("gn_g", C.c_void_p), ("gn_b", C.c_void_p),
                ("conv", Linear * 8),
                ("ln_g", C.c_void_p), ("ln_b", C.c_void_p),
                ("proj", Linear),
                ("dim", C.c_int32), ("heads", C.c_int32), ("pos_k", C.c_int32), ("pos_groups", C.c_int32),
                ("pos", C.POINTER(Linear)),
                ("enc_ln_g", C.c_void_p), ("enc_ln_b", C.c_void_p),
                ("n_layers", C.c_int32), ("layers", C.POINTER(HubertLayer))]


# name -> (restype, argtypes); must list every symbol of include/covomix_hip.h
SIGNATURES = {
    "cvx_version": (C.c_int, []),
    "cvx_hifigan_conv1d_f16x3": (C.c_int, [C.POINTER(Conv16Args), C.c_void_p]),
    "cvx_hifigan_conv_transpose1d_f16x3": (C.c_int, [C.POINTER(ConvT16Args), C.c_void_p]),
    "cvx_hifigan_split_channels_last": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "cvx_hifigan_post_channels_last_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                                     C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "cvx_hifigan_to_channels_last": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                               C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "cvx_hifigan_from_channels_last": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                 C.c_int32, C.c_void_p]),
    "cvx_mel_magnitude_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "cvx_mel_log_transpose_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "cvx_t2s_decode_steps": (C.c_int, [C.POINTER(T2SDecoder), C.c_int32, C.c_void_p]),
    "cvx_geglu_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p]),
    "cvx_hubert_conv0_workspace_floats": (C.c_int64, [C.c_int64, C.c_int32]),
    "cvx_hubert_conv0_gn_gelu_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                               C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "cvx_layernorm_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "cvx_hubert_group_pack_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "cvx_kmeans_argmin_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                        C.c_int32, C.c_void_p]),
    "cvx_resample_fir_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_int64, C.c_void_p]),
    "cvx_hubert_frames": (C.c_int32, [C.POINTER(HubertModel), C.c_int64]),
    "cvx_hubert_workspace_bytes": (C.c_int64, [C.POINTER(HubertModel), C.c_int64]),
    "cvx_hubert_extract_features": (C.c_int, [C.POINTER(HubertModel), C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                              C.c_int64, C.c_void_p]),
    "cvx_rope_attention_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_float, C.c_void_p, C.c_void_p]),
    "cvx_hifigan_convt_f32": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "cvx_hifigan_resblock_f32": (C.c_int, [C.POINTER(ResblockArgs), C.c_void_p]),
    "cvx_hifigan_resblock_f16x3": (C.c_int, [C.POINTER(Resblock16Args), C.c_void_p]),
    "cvx_hifigan_resblock_stage_f16x3": (C.c_int, [C.POINTER(Resblock16Args), C.c_int32, C.c_void_p]),
    "cvx_hifigan_conv1d_group_f16x3": (C.c_int, [C.POINTER(Conv16Args), C.c_int32, C.c_void_p]),
    "cvx_hifigan_resblock_pair_f16x3": (C.c_int, [C.POINTER(Respair16Args), C.c_void_p]),
    "cvx_hifigan_conv_transpose1d_f32": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p, C.c_void_p]),
    "cvx_hifigan_conv_transpose1d_packed_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "cvx_hifigan_pack_conv_transpose1d_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "cvx_pow2_scale_from_amax_f32": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "cvx_hifigan_pre_post_f32": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "cvx_last_error_string": (C.c_char_p, []),
    "cvx_gemm_bias_act_f32": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    "cvx_split_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "cvx_gemm_f16x3_workspace_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "cvx_rope_attention_workspace_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "cvx_hifigan_to_channels_last_scaled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                                      C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "cvx_amax_pow2_scale_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvx_split_f16_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "cvx_split_f16_colscale_il": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32,
                                            C.c_float, C.c_void_p, C.c_void_p]),
    "cvx_rownorm_scale_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "cvx_adarmsnorm_scaled_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                            C.c_int32, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "cvx_attention_f16x3_scaled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "cvx_gemm_f16x3": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p, C.c_void_p, C.c_float, C.POINTER(GemmSplitIO),
                                 C.c_void_p]),
    "cvx_gemm_f16x3_norm": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p, C.c_void_p, C.c_float, C.POINTER(GemmSplitIO),
                                      C.POINTER(GemmNorm), C.c_void_p]),
    "cvx_adarmsnorm_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                     C.c_int64, C.c_float, C.c_float, C.c_void_p]),
    "cvx_attention_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                    C.c_void_p]),
    "cvx_attention_f16x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "cvx_dwconv31_gelu_res_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_void_p]),
    "cvx_cfg_combine_axpy_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "cvx_embed_gather_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "cvx_time_fourier_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "cvx_hifigan_conv1d_f32": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "cvx_hifigan_packed_weight_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "cvx_hifigan_pack_weight_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "cvx_hifigan_post_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_float, C.c_void_p]),
    "cvx_wav_to_int16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "cvx_saturation_flag_reset": (C.c_int, [C.c_void_p]),
    "cvx_saturation_flag_query": (C.c_int, [C.POINTER(C.c_uint32), C.c_int32, C.c_void_p]),
    "cvx_stream_create_cu_mask": (C.c_int, [C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_void_p)]),
    "cvx_stream_destroy": (C.c_int, [C.c_void_p]),
    "cvx_clock_stamps": (C.c_int, [C.c_void_p, C.c_void_p]),
    # ragged batches (cu_seqlens)
    "cvx_attention_varlen_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_float, C.c_void_p]),
    "cvx_attention_f16x3_varlen": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_float,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvx_gemm_skinny_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_void_p]),
    "cvx_dwconv31_gelu_res_varlen_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                   C.c_int32, C.c_int32, C.c_void_p]),
}

_lib = None


class CovomixHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises loudly if it is missing - there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise CovomixHipError(
            f"{LIB_PATH} not found: the gfx950 HIP library has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "covomix_amd has no CPU / PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.cvx_version.restype = C.c_int
    if lib.cvx_version() != ABI_VERSION:          # struct layouts below are this version's (the C structs carry no size field)
        raise CovomixHipError(f"{LIB_PATH} reports ABI version {lib.cvx_version()}, this binding is written for {ABI_VERSION}: "
                              "rebuild the library (`python -c 'import __graft_entry__ as g; g.build()'`)")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI drift; let it propagate
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_DEBUG_SYNC = os.environ.get("CVX_DEBUG_SYNC") == "1"      # dev: synchronise after every call so a GPU fault names its kernel


def check(rc: int, what: str = "") -> None:
    if _DEBUG_SYNC:
        import sys
        import torch
        print("[cvx]", what, file=sys.stderr, flush=True)
        torch.cuda.synchronize()
    if rc != 0:
        msg = load().cvx_last_error_string()
        raise CovomixHipError(f"{what or 'covomix_hip'} failed (rc={rc}): {msg.decode() if msg else ''}")
