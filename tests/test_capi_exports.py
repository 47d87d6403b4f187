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
    assert lib.locov_abi_version() == _lib.ABI_VERSION == 6        # 2: range-guard word on the split entry points; 3: amax_out slots; 4: locov_zero_if_raised, exact fused workspaces, timing_read_ex; 5: locov_box_reg_loss, locov_grounding_ce_fwd / _bwd; 6: locov_res5_weight_prep, locov_gemm_segmean_supported


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


def test_cpu_tensors_are_rejected_loudly():
    import torch
    from locov_amd import ops
    from locov_amd._lib import LocovError
    with pytest.raises(LocovError, match="no CPU fallback"):
        ops.roi_align(torch.zeros(1, 4, 8, 8), torch.zeros(1, 5), 7, 1 / 16)
    with pytest.raises(LocovError):
        ops.linear(torch.zeros(4, 8), torch.zeros(3, 8))
