"""ctypes binding of ``libposetraj_hip.so`` (C ABI declared in ``include/posetraj_hip.h``).

There is no fallback: if the library is missing or a call fails, a ``RuntimeError`` is raised.  ``build()``
compiles the HIP sources for gfx950 in-tree (``posetraj_amd/libposetraj_hip.so``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.abspath(os.environ["PT_LIB"]) if os.environ.get("PT_LIB") else os.path.join(HERE, "libposetraj_hip.so")   # PT_LIB: A/B against another build on one box
SOURCES = ["api.hip", "igemm.hip", "ffn.hip", "lnlin.hip", "norm.hip", "attn.hip", "attn_general.hip", "elementwise.hip", "vae.hip", "vae_f32.hip", "clip.hip", "raster.hip", "train.hip", "gemm.hip", "backward.hip", "attn_bwd.hip"]
HEADERS = ["pt_common.h", "igemm_tail.h"]
ABI_VERSION = 8

_lib = None


class IgemmParams(C.Structure):
    """Mirror of ``pt_igemm_params`` (include/posetraj_hip.h)."""
    _fields_ = [
        ("x0", C.c_void_p), ("x1", C.c_void_p),
        ("C0", C.c_int32), ("C1", C.c_int32), ("ld0", C.c_int32), ("ld1", C.c_int32),
        ("Nimg", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad_h", C.c_int32), ("pad_w", C.c_int32),
        ("upsample2x", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("Kpad", C.c_int32),
        ("w", C.c_void_p), ("bias", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int32),
        ("res", C.c_void_p), ("ldr", C.c_int32),
        ("vec", C.c_void_p), ("ldv", C.c_int32), ("vec_mode", C.c_int32), ("vG", C.c_int32), ("vFS", C.c_int32),
        ("vS", C.c_int32), ("vB", C.c_int32),
        ("blend", C.c_void_p), ("ldb", C.c_int32), ("alpha", C.c_float),
        ("out_scale", C.c_float), ("act", C.c_int32), ("res_post", C.c_int32), ("out_f32", C.c_int32),
        ("cs_cols", C.c_int32), ("cs_scale", C.c_float),
        ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_int64),
        ("res_lo", C.c_void_p), ("out_lo", C.c_void_p),
    ]


class FfnParams(C.Structure):
    """Mirror of ``pt_ffn_params`` (include/posetraj_hip.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int32),
        ("M", C.c_int32), ("C", C.c_int32), ("inner", C.c_int32),
        ("w1", C.c_void_p), ("b1", C.c_void_p), ("kpad1", C.c_int32),
        ("w2", C.c_void_p), ("b2", C.c_void_p), ("kpad2", C.c_int32),
        ("out", C.c_void_p), ("ldo", C.c_int32),
        ("res", C.c_void_p), ("ldr", C.c_int32),
        ("vec", C.c_void_p), ("ldv", C.c_int32), ("vec_mode", C.c_int32), ("vG", C.c_int32), ("vFS", C.c_int32),
        ("vS", C.c_int32), ("vB", C.c_int32),
        ("blend", C.c_void_p), ("ldb", C.c_int32), ("alpha", C.c_float),
        ("pre_w", C.c_void_p), ("pre_b", C.c_void_p), ("pre_kpad", C.c_int32),
        ("pre_res", C.c_void_p), ("pre_ldr", C.c_int32),
        ("pre_vec", C.c_void_p), ("pre_ldv", C.c_int32), ("pre_vec_mode", C.c_int32), ("pre_vG", C.c_int32), ("pre_vFS", C.c_int32),
        ("pre_vS", C.c_int32), ("pre_vB", C.c_int32),
        ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float),
    ]


class LnLinParams(C.Structure):
    """Mirror of ``pt_lnlin_params`` (include/posetraj_hip.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("w", C.c_void_p), ("kpad", C.c_int32),
        ("bias", C.c_void_p),
        ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float),
        ("out", C.c_void_p), ("ldo", C.c_int32),
        ("cs_cols", C.c_int32), ("cs_scale", C.c_float),
    ]


class ConvF32Params(C.Structure):
    """Mirror of ``pt_conv_f32_params`` (include/posetraj_hip.h)."""
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p), ("out", C.c_void_p)] + \
               [(n, C.c_int32) for n in ("Nimg", "Hin", "Win", "Hout", "Wout", "Ci", "Co", "KH", "KW", "stride", "pad_h", "pad_w",
                                         "ldx", "ldw", "ldo", "ldr")] + [("scale", C.c_float)]


class GemmParams(C.Structure):
    """Mirror of ``pt_gemm_params`` (include/posetraj_hip.h)."""
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32),
        ("K", C.c_int64),
        ("sa_m", C.c_int64), ("sa_k", C.c_int64), ("sb_k", C.c_int64), ("sb_n", C.c_int64), ("sc_m", C.c_int64), ("sc_n", C.c_int64),
        ("nb0", C.c_int32), ("nb1", C.c_int32), ("nb2", C.c_int32), ("out_mode", C.c_int32),
        ("ba0", C.c_int64), ("ba1", C.c_int64), ("ba2", C.c_int64), ("bb0", C.c_int64), ("bb1", C.c_int64), ("bb2", C.c_int64),
        ("bc0", C.c_int64), ("bc1", C.c_int64), ("bc2", C.c_int64),
        ("alpha", C.c_float),
        ("splits", C.c_int32),
        ("g_H", C.c_int32), ("g_W", C.c_int32), ("g_OH", C.c_int32), ("g_OW", C.c_int32), ("g_KH", C.c_int32), ("g_KW", C.c_int32),
        ("g_stride", C.c_int32), ("g_pad_h", C.c_int32), ("g_pad_w", C.c_int32), ("g_reserved", C.c_int32),
        ("g_ld", C.c_int64),
    ]


# name -> (restype, argtypes); every symbol include/posetraj_hip.h declares
SIGNATURES = {
    "pt_abi_version": (C.c_int, []),
    "pt_last_error": (C.c_char_p, []),
    "pt_set_zero_page": (C.c_int, [C.c_void_p]),
    "pt_igemm_f16": (C.c_int, [C.POINTER(IgemmParams), C.c_void_p]),
    "pt_igemm_splitk_ws_bytes": (C.c_int64, [C.POINTER(IgemmParams)]),
    "pt_ffn_geglu_f16": (C.c_int, [C.POINTER(FfnParams), C.c_void_p]),
    "pt_ln_linear_f16": (C.c_int, [C.POINTER(LnLinParams), C.c_void_p]),
    "pt_ln_linear_set_ablation": (C.c_int, [C.c_int32]),
    "pt_conv2d_f32": (C.c_int, [C.POINTER(ConvF32Params), C.c_void_p]),
    "pt_groupnorm_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "pt_softmax_rows_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_float, C.c_void_p]),
    "pt_igemm_force_config": (C.c_int, [C.c_int32]),
    "pt_igemm_set_stamps": (C.c_int, [C.c_void_p, C.c_int64]),
    "pt_groupnorm_scratch_floats": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "pt_groupnorm_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                                     C.c_void_p, C.c_void_p]),
    "pt_groupnorm_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_float,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "pt_layernorm_f16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "pt_attn_spatial_f16": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "pt_attn_spatial_set_nqb": (C.c_int, [C.c_int32]),
    "pt_attn_temporal_f16": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "pt_attn_f16": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                              C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "pt_vae_time_conv_out": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "pt_frames_postprocess": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "pt_nhwc_to_nchw_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "pt_gaussian_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "pt_patchify_f16": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                  C.c_void_p]),
    "pt_act_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "pt_rasterize_tracks": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_void_p, C.c_void_p]),
    "pt_edm_train_input": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "pt_edm_loss": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64,
                              C.c_void_p, C.c_void_p]),
    "pt_gemm_f16": (C.c_int, [C.POINTER(GemmParams), C.c_void_p]),
    "pt_attn_fwd_lse_f16": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "pt_attn_bwd_f16": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                  C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "pt_attn_temporal_bwd_f16": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "pt_groupnorm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_void_p,
                                   C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pt_layernorm_bwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "pt_colsum_f16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pt_softmax_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "pt_softmax_bwd_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "pt_geglu_f16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "pt_geglu_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "pt_silu_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pt_lerp_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p, C.c_void_p]),
    "pt_dot_diff": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "pt_lerp_f16_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pt_dot_diff_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pt_scale_f16_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "pt_sigmoid_gather_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "pt_add_rowvec_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "pt_sumpool2x_f16": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pt_zero_insert2x_f16": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pt_edm_loss_bwd": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64,
                                  C.c_float, C.c_void_p, C.c_void_p]),
    "pt_adamw_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                               C.c_float, C.c_int32, C.c_float, C.c_void_p]),
    "pt_adamw_fused_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_int32, C.c_float, C.c_void_p, C.c_int32, C.c_void_p]),
    "pt_sumsq_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pt_pack_weight_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_void_p]),
    "pt_gemv_f16": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                              C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "pt_axpy_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "pt_silu_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "pt_timestep_embedding": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pt_nchw_to_nhwc_f16": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_void_p]),
    "pt_nhwc_to_nchw": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                  C.c_int32, C.c_void_p]),
    "pt_concat_camera": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p,
                                   C.c_void_p]),
    "pt_scale_concat_input": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_void_p]),
    "pt_cfg_euler_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pt_scale": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "pt_euler_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_int64,
                                C.c_void_p]),
    "pt_add_noise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "pt_resize_antialias_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                          C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pt_prof_enable": (C.c_int, [C.c_int32]),
    "pt_prof_collect": (C.c_int, [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pt_prof_collect_list": (C.c_int64, [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64]),
}


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> posetraj_amd/libposetraj_hip.so (cross-compiles without a GPU).  One object per
    source under ``csrc/_obj`` (re-compiled only when the source or a header is newer; up to 8 compiles in parallel, largest file first), then
    one link.  Refuses to run while ``PT_LIB`` points the loader at another library (an A/B build must not be overwritten
    by a build of the current tree)."""
    if os.environ.get("PT_LIB"):
        raise RuntimeError("posetraj_amd.hip.build: PT_LIB is set (A/B against another library); unset it to build the tree's own")
    from concurrent.futures import ThreadPoolExecutor
    headers = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(HERE, "..", "include", "posetraj_hip.h")]
    hdr_time = max(os.path.getmtime(h) for h in headers)
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    jobs = []
    for name in SOURCES:
        src, obj = os.path.join(CSRC, name), os.path.join(objdir, name + ".o")
        rem = obj + ".remarks"
        if force or not (os.path.exists(obj) and os.path.exists(rem)) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append((src, obj, rem))

    def compile_one(job):
        src, obj, rem = job
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed ({r.returncode}) on {os.path.basename(src)}:\n{r.stderr[-4000:]}")
        keep, on = [], False                                 # warnings are not remarks: show them (with their source excerpt)
        for ln in r.stderr.splitlines():
            if "warning:" in ln or "error:" in ln:
                on = True
            elif "remark:" in ln:
                on = False
            if on:
                keep.append(ln)
        if keep:
            print("\n".join(keep))
        with open(rem, "w") as f:
            f.write(r.stderr)

    if jobs:
        jobs.sort(key=lambda j: -os.path.getsize(j[0]))       # the long pole (igemm.hip: ~70 s of the build) starts first
        with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, 8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(objdir, name + ".o") for name in SOURCES]
    if jobs or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(o) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc link failed ({r.returncode}):\n{r.stderr[-4000:]}")
        _write_resource_report("\n".join(open(o + ".remarks").read() for o in objs))
    return LIB_PATH


RESOURCES_PATH = os.path.join(HERE, "build_resources.json")


def _write_resource_report(remarks: str) -> None:
    """Register / spill counts per kernel from hipcc's kernel-resource-usage remarks -> posetraj_amd/build_resources.json.
    The pipelined igemm kernels live at the 256-register limit: a spill inside their K loop is a reload behind the LDS-DMA
    queue (tests/test_host_cpu.py asserts they have none)."""
    import json
    import re
    out, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z ]*?)(?: \[[^\]]*\])?:\s+(\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    if out:
        with open(RESOURCES_PATH, "w") as f:
            json.dump(out, f, indent=0, sort_keys=True)


def source_digest() -> str:
    """sha256 over the HIP sources + headers of the library: identifies a BUILD independently of where it was compiled.
    Measurement files under profiles/ that only hold for one build (PMC traffic summaries) record it, and bench.py refuses
    to replay them for another."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted([os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.join(HERE, "..", "include", "posetraj_hip.h")]):
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()


def lib():
    """The loaded library; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(posetraj_amd has no CPU or PyTorch fallback path)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)            # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        if L.pt_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libposetraj_hip.so ABI {L.pt_abi_version()} != expected {ABI_VERSION}; rebuild")
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        raise RuntimeError(f"libposetraj_hip: {what} failed ({rc}): {lib().pt_last_error().decode()}")
