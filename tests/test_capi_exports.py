"""CPU-side checks of the drop-in boundary: the library builds for gfx950, loads, and exports
every symbol include/locov_hip.h declares (no compute calls: there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "locov_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(locov_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from locov_amd import build, _lib
    build.build_extension()
    return _lib.load()


def test_header_and_binding_agree(lib):
    from locov_amd import _lib
    names = _declared()
    assert len(names) >= 14
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)


def test_every_symbol_is_exported(lib):
    for name in _declared():
        assert hasattr(lib, name), name
    from locov_amd import _lib
    assert lib.locov_abi_version() == _lib.ABI_VERSION == 8        # 2: range-guard word on the split entry points; 3: amax_out slots; 4: locov_zero_if_raised, exact fused workspaces, timing_read_ex; 5: locov_box_reg_loss, locov_grounding_ce_fwd / _bwd; 6: locov_res5_weight_prep, locov_gemm_segmean_supported; 7: locov_sample_proposals, locov_gemm_tn_f32_split_b, locov_winograd_wgrad_f32_split_v; 8: locov_detect_postprocess


def test_argument_errors_are_reported_without_a_gpu(lib):
    from locov_amd import _lib
    # argument validation happens before any HIP call
    rc = lib.locov_gemm_nt_f32(None, 6, None, None, None, None, None, 4, 8, 4, 6, 0, None)
    assert rc == -1 and b"null pointer" in lib.locov_last_error()
    rc = lib.locov_rownorm_fwd(None, 4, 8, 7, 1e-12, None, None)
    assert rc == -1 and b"unknown mode" in lib.locov_last_error()
    rc = lib.locov_level_assign(None, -1, 2, 5, 224, 4, None, None)
    assert rc == -1
    with pytest.raises(_lib.LocovError):
        _lib.check(rc, "locov_level_assign")
    # empty inputs are a no-op success
    assert lib.locov_roi_align_fwd(None, 1, 4, 8, 8, None, 0, 7, 7, 0.0625, 0, 1, None, None) == 0


def test_round5_entry_points_validate_their_arguments(lib):
    """locov_res5_weight_prep / locov_amax_bound / locov_gemm_segmean_supported: argument errors before any HIP call."""
    import ctypes
    from locov_amd import _lib
    assert lib.locov_res5_weight_prep(None, 0, None, None) == 0                       # no jobs: a no-op
    assert lib.locov_res5_weight_prep(None, 3, None, None) == -1 and b"null job list" in lib.locov_last_error()
    jobs = (_lib.WeightPrepJob * 1)()
    assert lib.locov_res5_weight_prep(jobs, _lib.WEIGHT_PREP_MAX_JOBS + 1, None, None) == -1
    jobs[0].w, jobs[0].out, jobs[0].scale, jobs[0].kind, jobs[0].N, jobs[0].K = 256, 512, 1.0, _lib.PREP_PLAIN, 64, 48
    assert lib.locov_res5_weight_prep(jobs, 1, None, None) == -1 and b"K % 32" in lib.locov_last_error()
    jobs[0].K, jobs[0].kind = 64, 17
    assert lib.locov_res5_weight_prep(jobs, 1, None, None) == -1 and b"unknown kind" in lib.locov_last_error()
    jobs[0].kind, jobs[0].out = _lib.PREP_TRANSPOSE, 513
    assert lib.locov_res5_weight_prep(jobs, 1, None, None) == -1 and b"misaligned" in lib.locov_last_error()
    assert lib.locov_amax_bound(None, None, None, 0, None, None) == 0
    assert lib.locov_amax_bound(None, None, None, _lib.AMAX_BOUND_MAX + 1, None, None) == -1
    ptrs, ns, ms = (ctypes.c_void_p * 1)(256), (ctypes.c_int64 * 1)(6), (ctypes.c_float * 1)(1.0)
    assert lib.locov_amax_bound(ptrs, ns, ms, 1, ctypes.c_void_p(512), None) == -1 and b"numel % 4" in lib.locov_last_error()
    # the mean-fused convolution: the 256x256 form takes any M (pre-split x, ROI-major residual, >= 1 024 tiles), the 128x128 form M * N * 4 < 2^32
    big = _lib.GEMM_A_SPLIT | _lib.SEGMEAN_RES_ROI_MAJOR | _lib.EPI_RELU
    assert lib.locov_gemm_segmean_supported(512, 49 * 12000, 2048, 512, 49, big) == 1
    assert lib.locov_gemm_segmean_supported(512, 49 * 12000, 2048, 512, 49, _lib.EPI_RELU) == 0         # fp32 x: the 128x128 form, 4.8 GB
    assert lib.locov_gemm_segmean_supported(512, 49 * 1000, 2048, 512, 49, _lib.EPI_RELU) == 1
    assert lib.locov_gemm_segmean_supported(512, 49 * 1000, 2048, 512, 40, big) == 0                   # seg < 43


def test_sample_proposals_validates_its_arguments(lib):
    """locov_sample_proposals: argument errors before any HIP call."""
    import ctypes
    from locov_amd import _lib
    none = [None] * 8
    outs = [None] * 7
    off = (ctypes.c_int * 2)(0, 100)
    assert lib.locov_sample_proposals(*none, off, off, 0, 16, 4, 80, *outs, None) == 0                  # no images: a no-op
    assert lib.locov_sample_proposals(*none, off, off, 1, 0, 0, 80, *outs, None) == 0                   # no budget: a no-op
    assert lib.locov_sample_proposals(*none, off, off, 1, 16, 17, 80, *outs, None) == -1 and b"max_pos <= budget" in lib.locov_last_error()
    assert lib.locov_sample_proposals(*none, off, off, _lib.LABEL_MAX_IMAGES + 1, 16, 4, 80, *outs, None) == -1
    big = (ctypes.c_int * 2)(0, _lib.SAMPLE_MAX_PROPOSALS + 1)
    assert lib.locov_sample_proposals(*none, big, off, 1, 16, 4, 80, *outs, None) == -1 and b"proposals per image" in lib.locov_last_error()
    assert lib.locov_sample_proposals(*none, off, off, 1, 16, 4, 80, *outs, None) == -1 and b"null pointer" in lib.locov_last_error()


def test_detect_postprocess_validates_its_arguments(lib):
    """locov_detect_postprocess: argument errors before any HIP call."""
    import ctypes
    from locov_amd import _lib
    off, hw = (ctypes.c_int * 2)(0, 100), (ctypes.c_float * 2)(800, 1333)
    tail = [None, 0] + [None] * 6
    call = lambda n_img, k, topk, offsets=off: lib.locov_detect_postprocess(None, 1204, k, None, None, offsets, hw, n_img, 10, 10, 5, 5, 4.1, 0.05, 0.5, topk, *tail)
    assert call(0, 1203, 100) == 0                                                                     # no images: a no-op
    assert call(_lib.LABEL_MAX_IMAGES + 1, 1203, 100) == -1
    assert call(1, 1 << 15, 100) == -1 and b"classes" in lib.locov_last_error()
    assert call(1, 1203, 0) == -1 and b"topk" in lib.locov_last_error()
    assert call(1, 1203, _lib.DETECT_MAX_CANDIDATES + 1) == -1
    assert call(1, 1203, 100, (ctypes.c_int * 2)(0, 1 << 14)) == -1 and b"proposals per image" in lib.locov_last_error()
    assert call(1, 1203, 100) == -1 and b"null pointer" in lib.locov_last_error()
    assert lib.locov_detect_postprocess_workspace_bytes(0, 1) == 0 and lib.locov_detect_postprocess_workspace_bytes(1000, 1) >= 1000 * 24 + 8192 * 8


def test_presplit_weight_gradient_entry_points_validate_their_arguments(lib):
    """locov_gemm_tn_f32_split_b / locov_winograd_wgrad_f32_split_v: argument errors before any HIP call."""
    import ctypes
    p = ctypes.c_void_p
    assert lib.locov_gemm_tn_f32_split_b(None, 64, 0, None, 64, 0, None, 64, 0, 100, 64, 64, 1, None, None, 16.0, None, None, 0, None) == -1
    assert b"null pointer" in lib.locov_last_error()
    assert lib.locov_gemm_tn_f32_split_b(p(256), 64, 0, p(512), 32, 0, p(1024), 64, 0, 100, 64, 64, 1, None, p(2048), 16.0, None, p(4096), 1 << 30, None) == -1
    assert b"ldb < K" in lib.locov_last_error()
    assert lib.locov_gemm_tn_f32_split_b(p(256), 64, 0, p(512), 68, 0, p(1024), 64, 0, 100, 64, 60, 1, None, p(2048), 16.0, None, p(4096), 1 << 30, None) < 0
    assert b"multiples of 8" in lib.locov_last_error()
    assert lib.locov_winograd_wgrad_f32_split_v(None, p(256), 10, 64, 64, 0, None, p(512), None, p(1024), 1 << 30, None) == -1
    assert b"null transformed input" in lib.locov_last_error()
    assert lib.locov_winograd_wgrad_f32_split_v(p(256), p(512), 10, 36, 64, 0, None, p(1024), None, p(2048), 1 << 30, None) == -1
    assert b"multiple of 8" in lib.locov_last_error()


def test_cpu_tensors_are_rejected_loudly():
    import torch
    from locov_amd import ops
    from locov_amd._lib import LocovError
    with pytest.raises(LocovError, match="no CPU fallback"):
        ops.roi_align(torch.zeros(1, 4, 8, 8), torch.zeros(1, 5), 7, 1 / 16)
    with pytest.raises(LocovError):
        ops.linear(torch.zeros(4, 8), torch.zeros(3, 8))
