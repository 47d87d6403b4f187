"""ctypes binding of liblocov_hip.so (the C ABI of include/locov_hip.h).

The library is loaded on first use and the import fails loudly if it is missing or lacks a
symbol: there is no PyTorch / CPU fallback for the hot path.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint, c_void_p, POINTER

HERE = os.path.dirname(os.path.abspath(__file__))
# LOCOV_HIP_LIB: developer override to A/B an experimental build of the same ABI (tools/)
LIB_PATH = os.environ.get("LOCOV_HIP_LIB") or os.path.join(HERE, "liblocov_hip.so")

OK = 0
F32, BF16 = 0, 1
NORM_NONE, NORM_L2, NORM_STANDARDIZE = 0, 1, 2
EPI_RELU = 1
WINO_OUT_ROI_MAJOR = 0x100
WINO_IN_ROI_MAJOR = 0x200
SEGMEAN_RES_ROI_MAJOR = 0x400
GEMM_A_SPLIT = 0x1000
EPI_OUT_SPLIT = 0x2000
EPI_RES_SPLIT = 0x4000
MAX_LEVELS = 8
ZERO_LIST_MAX = 24
LABEL_MAX_IMAGES, LABEL_MAX_THRESHOLDS, SAMPLE_MAX_PROPOSALS = 64, 6, 4096
ABI_VERSION = 8
DETECT_MAX_CANDIDATES, DETECT_FLAG_NONFINITE, DETECT_FLAG_OVERFLOW = 8192, 1, 2

_p = c_void_p  # device pointer

WEIGHT_PREP_MAX_JOBS = 32
AMAX_BOUND_MAX = 4
PREP_PLAIN, PREP_TRANSPOSE, PREP_IM2COL, PREP_IM2COL_FLIP, PREP_WINO, PREP_WINO_FLIP = range(6)


class WeightPrepJob(ctypes.Structure):
    """locov_weight_prep_job of include/locov_hip.h"""
    _fields_ = [("w", c_void_p), ("row_scale", c_void_p), ("out", c_void_p), ("scale", c_float), ("kind", c_int), ("N", c_int), ("K", c_int)]


# name -> (restype, argtypes); mirrors include/locov_hip.h declaration by declaration
SIGNATURES = {
    "locov_abi_version": (c_int, []),
    "locov_last_error": (c_char_p, []),
    "locov_launch_count": (c_int64, []),
    "locov_device_info": (c_int, [POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "locov_level_assign": (c_int, [_p, c_int64, c_int, c_int, c_int, c_int, _p, _p]),
    "locov_roi_align_fwd": (c_int, [_p, c_int, c_int, c_int, c_int, _p, c_int64, c_int, c_int, c_float,
                                    c_int, c_int, _p, _p]),
    "locov_roi_align_bwd": (c_int, [_p, c_int, c_int, c_int, c_int, _p, c_int64, c_int, c_int, c_float,
                                    c_int, c_int, _p, _p]),
    "locov_roi_align_levels_fwd": (c_int, [POINTER(c_void_p), POINTER(c_int), POINTER(c_int), POINTER(c_float),
                                           c_int, c_int, c_int, _p, _p, c_int64, c_int, c_int, c_int, c_int,
                                           _p, _p]),
    "locov_nchw_to_nhwc": (c_int, [_p, c_int, c_int, c_int, c_int, _p, c_int, _p]),
    "locov_roi_align_nhwc_fwd": (c_int, [_p, c_int, c_int, c_int, c_int, c_int, _p, c_int64, c_int, c_int,
                                         c_float, c_int, c_int, c_int, c_int, _p, c_int, _p]),
    "locov_roi_align_nhwc_ld_fwd": (c_int, [_p, c_int, c_int, c_int, c_int, c_int, _p, c_int64, c_int, c_int,
                                            c_float, c_int, c_int, c_int, c_int, _p, c_int64, c_int, _p]),
    "locov_roi_align_nhwc_affine_fwd": (c_int, [_p, c_int, c_int, c_int, c_int, c_int, c_int64, _p, c_int64, c_int, c_int,
                                                c_float, c_int, c_int, c_int, c_int, _p, _p, c_int, _p, c_int64, c_int, _p]),
    "locov_roi_align_from_nhwc_fwd": (c_int, [_p, c_int, c_int, c_int, c_int, _p, c_int64, c_int, c_int, c_float,
                                              c_int, c_int, _p, _p]),
    "locov_roi_align_plan_bytes": (c_int64, [c_int64]),
    "locov_roi_align_from_nhwc_fwd_ex": (c_int, [_p, c_int, c_int, c_int, c_int, _p, c_int64, c_int, c_int, c_float,
                                                 c_int, c_int, c_int, _p, c_int64, _p, _p]),
    "locov_spatial_mean_fwd": (c_int, [_p, c_int64, c_int, c_int, c_int, _p, _p]),
    "locov_gemm_nt_f32": (c_int, [_p, c_int64, _p, _p, _p, _p, _p, c_int64, c_int64, c_int, c_int, c_uint, _p]),
    "locov_conv3x3_nhwc_f32": (c_int, [_p, c_int64, c_int, c_int, c_int, c_int, _p, _p, _p, _p, _p, c_int, c_uint,
                                       _p]),
    "locov_pack_conv3x3_weight": (c_int, [_p, c_int, c_int, _p, c_int, _p]),
    "locov_winograd_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "locov_winograd_pack_weight": (c_int, [_p, c_int, c_int, _p, _p]),
    "locov_winograd_conv3x3_f32": (c_int, [_p, c_int64, c_int, _p, _p, _p, _p, c_int64, c_int, c_uint, _p, c_int64, _p]),
    "locov_winograd_conv3x3_f32_split": (c_int, [_p, c_int64, c_int, _p, c_float, c_float, _p, _p, _p, c_int64, c_int, c_uint,
                                                 _p, c_int64, _p, _p]),
    "locov_gemm_nt_batched_f32": (c_int, [_p, c_int64, c_int64, _p, c_int64, _p, c_int64, c_int64, c_int64, c_int,
                                          c_int, c_int, _p]),
    "locov_gemm_timing_enable": (c_int, [c_int]),
    "locov_gemm_timing_read": (c_int, [c_int, _p, _p, _p]),
    "locov_gemm_timing_read_ex": (c_int, [c_int, _p, _p, _p, _p]),
    "locov_frozen_bn_fold": (c_int, [_p, _p, _p, _p, c_float, c_int, _p, _p, _p]),
    "locov_label_proposals": (c_int, [_p, POINTER(c_int), _p, _p, POINTER(c_int), c_int, POINTER(c_float), POINTER(c_float),
                                      POINTER(c_int), c_int, c_int64, _p, _p, _p, _p, _p, _p, _p]),
    "locov_sample_proposals": (c_int, [_p, _p, _p, _p, _p, _p, _p, _p, POINTER(c_int), POINTER(c_int), c_int, c_int, c_int, c_int64,
                                       _p, _p, _p, _p, _p, _p, _p, _p]),
    "locov_box_reg_loss": (c_int, [_p, _p, _p, c_int64, _p, c_int64, c_int64, c_float, c_float, c_float, c_float, c_float, _p, _p, _p]),
    "locov_grounding_ce_fwd": (c_int, [_p, _p, _p, _p, c_int, c_int, c_int, _p, _p]),
    "locov_grounding_ce_bwd": (c_int, [_p, _p, _p, _p, c_int, c_int, c_int, _p, _p, _p, _p, _p, _p, _p]),
    "locov_nms_workspace_bytes": (c_int64, [c_int64]),
    "locov_nms_sorted": (c_int, [_p, c_int64, c_float, _p, _p, _p, _p]),
    "locov_detect_postprocess_workspace_bytes": (c_int64, [c_int64, c_int]),
    "locov_detect_postprocess": (c_int, [_p, c_int64, c_int, _p, _p, POINTER(c_int), POINTER(c_float), c_int, c_float, c_float, c_float,
                                         c_float, c_float, c_float, c_float, c_int, _p, c_int64, _p, _p, _p, _p, _p, _p]),
    "locov_grounding_fwd": (c_int, [_p, c_int, c_int, c_int, _p, _p, c_float, _p, _p, _p]),
    "locov_grounding_bwd": (c_int, [_p, c_int, c_int, c_int, _p, _p, c_float, _p, _p, _p, _p]),
    "locov_token_attention_fwd": (c_int, [_p, c_int64, c_int, _p, _p, c_int, c_int, c_float, c_int, c_int, _p, _p, _p, _p]),
    "locov_rownorm_fwd": (c_int, [_p, c_int64, c_int, c_int, c_float, _p, _p]),
    "locov_rownorm_bwd": (c_int, [_p, _p, c_int64, c_int, c_int, c_float, _p, _p]),
    "locov_pool_fc_bwd_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "locov_pool_fc_bwd": (c_int, [_p, c_int64, c_int, _p, c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p, c_int64, _p]),
    "locov_sim_gemm_bwd_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "locov_sim_gemm_bwd": (c_int, [_p, _p, _p, c_int64, c_int, c_int, _p, _p, _p, c_int64, _p]),
    "locov_f32_to_bf16": (c_int, [_p, c_int64, _p, _p]),
    "locov_gemm_nt_bf16": (c_int, [_p, c_int64, _p, _p, _p, _p, _p, c_int64, c_int64, c_int, c_int, c_uint, _p]),
    "locov_conv3x3_nhwc_bf16": (c_int, [_p, c_int64, c_int, c_int, c_int, c_int, _p, _p, _p, _p, _p, c_int, c_uint, _p]),
    "locov_split_f16x2_pack": (c_int, [_p, c_int64, c_int, c_int64, c_float, _p, _p, _p]),
    "locov_gemm_nt_f32_split": (c_int, [_p, c_int64, _p, _p, _p, _p, _p, c_int64, c_int64, c_int, c_int, c_uint, c_float,
                                        c_float, _p, _p]),
    "locov_gemm_segmean_supported": (c_int, [c_int64, c_int64, c_int, c_int, c_int, c_uint]),
    "locov_gemm_segmean_workspace_bytes": (c_int64, [c_int64, c_int]),
    "locov_gemm_nt_f32_split_segmean": (c_int, [_p, c_int64, _p, _p, _p, _p, _p, c_int64, c_int, c_int, c_int, c_uint, c_float,
                                                c_float, _p, c_int64, _p, _p]),
    "locov_gemm_nt_batched_f32_split": (c_int, [_p, c_int64, c_int64, _p, c_int64, _p, c_int64, c_int64, c_int64, c_int,
                                                c_int, c_int, c_float, c_float, _p, _p]),
    "locov_sim_gemm_bf16": (c_int, [_p, _p, c_int64, c_int, c_int, _p, c_int64, _p]),
    "locov_gemm_nt_f32_ex": (c_int, [_p, c_int64, _p, c_int64, _p, _p, _p, _p, _p, c_int64, c_int64, c_int, c_int, c_uint, _p]),
    "locov_conv3x3_nhwc_f32_ex": (c_int, [_p, c_int64, c_int, c_int, c_int, c_int, _p, _p, _p, _p, _p, _p, c_int, c_uint, _p]),
    "locov_winograd_conv3x3_f32_ex": (c_int, [_p, c_int64, c_int, _p, _p, _p, _p, _p, c_int64, c_int, c_uint, _p, c_int64, _p]),
    "locov_gemm_tn_workspace_bytes": (c_int64, [c_int64, c_int, c_int, c_int]),
    "locov_gemm_tn_f32": (c_int, [_p, c_int64, c_int64, _p, c_int64, c_int64, _p, c_int64, c_int64, c_int64, c_int, c_int, c_int,
                                  _p, _p, c_int64, _p]),
    "locov_split_scale_from_amax": (c_int, [_p, c_int64, c_float, _p, _p]),
    "locov_amax_bound": (c_int, [POINTER(c_void_p), POINTER(c_int64), POINTER(c_float), c_int, _p, _p]),
    "locov_split_scale_from_amax_zeroed": (c_int, [_p, c_int64, c_float, _p, _p]),
    "locov_gemm_nt_f32_split_ex": (c_int, [_p, c_int64, _p, _p, _p, _p, _p, _p, c_int64, c_int64, c_int, c_int, c_uint, c_float, _p,
                                           c_float, _p, _p, _p]),
    "locov_gemm_tn_f32_split": (c_int, [_p, c_int64, c_int64, _p, c_int64, c_int64, _p, c_int64, c_int64, c_int64, c_int, c_int, c_int,
                                        _p, _p, c_float, _p, _p, c_int64, _p]),
    "locov_gemm_tn_f32_split_b": (c_int, [_p, c_int64, c_int64, _p, c_int64, c_int64, _p, c_int64, c_int64, c_int64, c_int, c_int, c_int,
                                        _p, _p, c_float, _p, _p, c_int64, _p]),
    "locov_winograd_conv3x3_f32_split_ex": (c_int, [_p, c_int64, c_int, _p, c_float, c_float, c_int, _p, _p, _p, _p, c_int64, c_int,
                                                    c_uint, c_float, _p, c_int64, _p, _p, _p]),
    "locov_winograd_wgrad_f32_split": (c_int, [_p, _p, c_int64, c_int, c_int, c_uint, _p, _p, _p, _p, c_int64, _p]),
    "locov_winograd_wgrad_f32_split_v": (c_int, [_p, _p, c_int64, c_int, c_int, c_uint, _p, _p, _p, _p, c_int64, _p]),
    "locov_conv1x1_winograd_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "locov_conv1x1_winograd_workspace_bytes_for": (c_int64, [c_int64, c_int, c_int, c_int, c_int64]),
    "locov_roi_align_winograd_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "locov_roi_align_winograd_conv3x3_f32_split": (c_int, [_p, c_int, c_int, c_int, c_int, c_int64, _p, c_int64, c_int, c_float, c_int, c_int,
                                                           _p, _p, _p, c_float, c_float, _p, _p, _p, c_int64, c_int, c_uint, c_float, _p, c_int64,
                                                           _p, _p]),
    "locov_conv1x1_winograd_conv3x3_f32_split": (c_int, [_p, c_int64, c_int, c_float, _p, c_float, _p, _p, c_int64, c_int, _p, c_float, c_float,
                                                         _p, _p, _p, c_int64, c_int, c_uint, c_float, _p, c_int64, _p, _p]),
    "locov_winograd_wgrad_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "locov_winograd_wgrad_f32": (c_int, [_p, _p, c_int64, c_int, c_int, c_uint, _p, _p, _p, c_int64, _p]),
    "locov_res5_weight_prep": (c_int, [POINTER(WeightPrepJob), c_int, _p, _p]),
    "locov_weight_transpose_scale": (c_int, [_p, c_int, c_int, _p, _p, _p]),
    "locov_conv3x3_weight_flip": (c_int, [_p, c_int, c_int, _p, _p, _p]),
    "locov_im2col3x3_nhwc": (c_int, [_p, c_int64, c_int, c_int, c_int, _p, _p]),
    "locov_conv3x3_wgrad_unpack": (c_int, [_p, c_int, c_int, _p, _p, _p]),
    "locov_relu_mask": (c_int, [_p, _p, c_int64, _p, _p, _p]),
    "locov_zero_if_raised": (c_int, [POINTER(c_void_p), POINTER(c_int64), c_int, _p, _p]),
    "locov_spatial_mean_bwd": (c_int, [_p, _p, c_int64, c_int, c_int, _p, _p, _p]),
    "locov_rows_stride2": (c_int, [_p, c_int, c_int, c_int, c_int, c_int, _p, _p]),
    "locov_roi_align_nhwc_bwd": (c_int, [_p, c_int64, c_int, c_int, c_int, c_int, _p, c_int64, c_int, c_int, c_float, c_int, c_int,
                                         c_int, c_int, _p, _p]),
    "locov_box_head_fwd": (c_int, [_p, c_int64, c_int, c_int, c_int, _p, _p, _p, _p, _p, _p, c_int, c_int,
                                   c_int, c_int, _p, _p, _p, _p, _p, _p]),
}

_lib = None


class LocovError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load liblocov_hip.so and bind every symbol of the header; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LocovError(
            f"{LIB_PATH} not found: build it with `python -m locov_amd.build` "
            "(or __graft_entry__.build()).  The LSM ROI-head path has no fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise LocovError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.locov_abi_version() != ABI_VERSION:
        raise LocovError(f"ABI version mismatch: library {lib.locov_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != OK:
        msg = load().locov_last_error()
        raise LocovError(f"{what or 'liblocov_hip'} failed (code {rc}): {msg.decode() if msg else ''}")
