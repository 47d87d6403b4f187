"""Tensor-level wrappers over the C ABI (include/locov_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every function below checks
its arguments, allocates the outputs and enqueues the hand-written gfx950 kernels on
torch's current stream.  Nothing in this module computes with torch ops, and there is no
CPU path: tensors must live on a ROCm device.
"""
from __future__ import annotations

import ctypes
import itertools
import math
import threading
from typing import NamedTuple, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import BF16, F32, NORM_L2, NORM_NONE, NORM_STANDARDIZE, LocovError, check

__all__ = [
    "level_assign", "roi_align", "roi_align_levels", "nchw_to_nhwc", "roi_align_nhwc", "spatial_mean",
    "linear", "rownorm", "to_bf16", "sim_gemm_bf16", "box_head", "NORM_NONE", "NORM_L2",
    "NORM_STANDARDIZE", "F32", "BF16",
]


def _dev(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise LocovError(f"{name} is on {t.device}: the LSM ROI-head kernels only run on a ROCm GPU "
                         "(there is no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _rows(t: torch.Tensor, name: str) -> torch.Tensor:
    """2-D fp32 device matrix whose rows may be a column block of a wider matrix (stride(1) == 1, any
    16-byte-aligned row stride); anything else is made contiguous."""
    if (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape[0] > 0
            and t.stride(1) == 1 and t.stride(0) >= t.shape[1] and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0):
        return t
    return _dev(t, name)


def _out(out: Optional[torch.Tensor], shape, ref: torch.Tensor, name: str) -> torch.Tensor:
    """The result tensor of an op: a fresh one, or the caller's `out` (a contiguous fp32 device tensor of that shape -- e.g. a
    row slice of a larger matrix: the training step keeps the rows of both Res5 calls in ONE matrix per activation)."""
    if out is None:
        return torch.empty(shape, dtype=torch.float32, device=ref.device)
    if tuple(out.shape) != tuple(shape) or out.dtype != torch.float32 or out.device != ref.device or not out.is_contiguous():
        raise ValueError(f"{name}: out must be a contiguous fp32 {tuple(shape)} tensor on {ref.device}")
    return out


def _stream(t: torch.Tensor) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def _dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {dt}")


# --------------------------------------------------------------------------------------
def level_assign(boxes: torch.Tensor, min_level: int, max_level: int, canonical_box_size: int = 224,
                 canonical_level: int = 4) -> torch.Tensor:
    """boxes [R,4] XYXY fp32 -> int64 [R] level index in [0, max_level-min_level]."""
    boxes = _dev(boxes, "boxes")
    if boxes.dim() != 2 or boxes.shape[1] != 4:
        raise ValueError(f"boxes must be [R,4], got {tuple(boxes.shape)}")
    out = torch.empty((boxes.shape[0],), dtype=torch.int64, device=boxes.device)
    with torch.cuda.device(boxes.device):
        check(_lib.load().locov_level_assign(_ptr(boxes), boxes.shape[0], int(min_level), int(max_level),
                                             int(canonical_box_size), int(canonical_level), _ptr(out),
                                             _stream(boxes)), "locov_level_assign")
    return out


class _ROIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, ph, pw, scale, sampling_ratio, aligned, mode=0):
        N, C, H, W = feat.shape
        R = rois.shape[0]
        out = torch.empty((R, C, ph, pw), dtype=torch.float32, device=feat.device)
        with torch.cuda.device(feat.device):
            if C % 4 == 0 and R > 0 and ph * pw <= 2048:
                # fast path (same bits): gather from a channels-last copy, transpose in LDS
                nhwc = torch.empty((N, H, W, C), dtype=torch.float32, device=feat.device)
                check(_lib.load().locov_nchw_to_nhwc(_ptr(feat), N, C, H, W, _ptr(nhwc), F32, _stream(feat)),
                      "locov_nchw_to_nhwc")
                lib = _lib.load()
                ws = _workspace("roi_plan", feat, int(lib.locov_roi_align_plan_bytes(R))) if mode else None
                check(lib.locov_roi_align_from_nhwc_fwd_ex(_ptr(nhwc), N, H, W, C, _ptr(rois), R, ph, pw, float(scale),
                                                           int(sampling_ratio), int(aligned), int(mode), _ptr(ws),
                                                           ws.numel() if ws is not None else 0, _ptr(out), _stream(feat)),
                      "locov_roi_align_from_nhwc_fwd_ex")
            else:
                check(_lib.load().locov_roi_align_fwd(_ptr(feat), N, C, H, W, _ptr(rois), R, ph, pw, float(scale),
                                                      int(sampling_ratio), int(aligned), _ptr(out), _stream(feat)),
                      "locov_roi_align_fwd")
        ctx.save_for_backward(rois)
        ctx.args = (N, C, H, W, ph, pw, scale, sampling_ratio, aligned)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (rois,) = ctx.saved_tensors
        N, C, H, W, ph, pw, scale, sampling_ratio, aligned = ctx.args
        grad_out = _dev(grad_out, "grad_out")
        gf = torch.zeros((N, C, H, W), dtype=torch.float32, device=grad_out.device)
        with torch.cuda.device(gf.device):
            check(_lib.load().locov_roi_align_bwd(_ptr(grad_out), N, C, H, W, _ptr(rois), rois.shape[0], ph, pw,
                                                  float(scale), int(sampling_ratio), int(aligned), _ptr(gf),
                                                  _stream(gf)), "locov_roi_align_bwd")
        return gf, None, None, None, None, None, None, None


ROIALIGN_MODES = {"exact": 0, "fast": 1}


def roi_align(feat: torch.Tensor, rois: torch.Tensor, output_size, spatial_scale: float,
              sampling_ratio: int = 0, aligned: bool = True, mode: str = "exact") -> torch.Tensor:
    """feat [N,C,H,W] fp32, rois [R,5] (batch_idx,x0,y0,x1,y1) -> [R,C,ph,pw] (differentiable in feat).
    mode "exact": torchvision's per-sample order, bit-identical to the CPU oracle; "fast": within 1e-5 of it -- separable
    per-pixel weights and the proposal's pixel window staged in LDS (include/locov_hip.h, LOCOV_ROIALIGN_FAST)."""
    feat = _dev(feat, "feat")
    rois = _dev(rois, "rois")
    if feat.dim() != 4:
        raise ValueError(f"feat must be [N,C,H,W], got {tuple(feat.shape)}")
    if rois.dim() != 2 or rois.shape[1] != 5:
        raise ValueError(f"rois must be [R,5], got {tuple(rois.shape)}")
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    return _ROIAlignFn.apply(feat, rois, int(ph), int(pw), float(spatial_scale), int(sampling_ratio), bool(aligned),
                             ROIALIGN_MODES[mode])


def roi_align_levels(feats: Sequence[torch.Tensor], scales: Sequence[float], rois: torch.Tensor,
                     levels: Optional[torch.Tensor], output_size: int, sampling_ratio: int = 0,
                     aligned: bool = True) -> torch.Tensor:
    """Multi-level pooler in one launch: ROI r reads feats[levels[r]]."""
    L = len(feats)
    if not (1 <= L <= _lib.MAX_LEVELS) or len(scales) != L:
        raise ValueError("need 1..8 feature levels with one scale each")
    feats = [_dev(f, f"feats[{i}]") for i, f in enumerate(feats)]
    rois = _dev(rois, "rois")
    N, C = feats[0].shape[:2]
    for f in feats:
        if f.shape[0] != N or f.shape[1] != C:
            raise ValueError("all levels must share N and C")
    if L > 1:
        levels = _dev(levels, "levels", torch.int64)
        if levels.shape[0] != rois.shape[0]:
            raise ValueError("levels must have one entry per ROI")
    R = rois.shape[0]
    out = torch.empty((R, C, output_size, output_size), dtype=torch.float32, device=rois.device)
    fp = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
    hh = (ctypes.c_int * L)(*[f.shape[2] for f in feats])
    ww = (ctypes.c_int * L)(*[f.shape[3] for f in feats])
    sc = (ctypes.c_float * L)(*[float(s) for s in scales])
    with torch.cuda.device(rois.device):
        check(_lib.load().locov_roi_align_levels_fwd(fp, hh, ww, sc, L, N, C, _ptr(rois),
                                                     _ptr(levels if L > 1 else None), R, output_size, output_size,
                                                     int(sampling_ratio), int(aligned), _ptr(out), _stream(rois)),
              "locov_roi_align_levels_fwd")
    return out


def nchw_to_nhwc(feat: torch.Tensor, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    feat = _dev(feat, "feat")
    N, C, H, W = feat.shape
    out = torch.empty((N, H, W, C), dtype=dtype, device=feat.device)
    with torch.cuda.device(feat.device):
        check(_lib.load().locov_nchw_to_nhwc(_ptr(feat), N, C, H, W, _ptr(out), _dtype_code(dtype), _stream(feat)),
              "locov_nchw_to_nhwc")
    return out


def roi_align_nhwc(feat: torch.Tensor, rois: torch.Tensor, output_size: int, spatial_scale: float,
                   sampling_ratio: int = 0, aligned: bool = True, bin_stride: int = 1,
                   out_dtype: torch.dtype = torch.float32, pos_major: bool = False,
                   out: Optional[torch.Tensor] = None, ch_scale: Optional[torch.Tensor] = None,
                   ch_shift: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    """feat [N,H,W,C] (fp32|bf16) -> [R, o, o, C] with o = ceil(P/bin_stride), or, with pos_major,
    [o, o, R, C] (position-major pixel rows, the fast layout of the Res5 GEMMs).
    feat may be a channel slice of a wider channels-last map (pixel stride > C).
    out: optional destination given as the pixel-row matrix [o*o*R, C]; its rows may be a column block
    of a wider matrix (row stride >= C).
    ch_scale / ch_shift [C], relu: per-channel affine + ReLU applied to the pooled values."""
    if not isinstance(feat, torch.Tensor) or feat.dim() != 4:
        raise ValueError("roi_align_nhwc: feat must be [N,H,W,C]")
    N, H, W, C = feat.shape
    fld = C
    if (feat.is_cuda and feat.stride(3) == 1 and feat.stride(2) > C and feat.stride(2) % 4 == 0
            and feat.stride(1) == W * feat.stride(2) and feat.stride(0) == H * W * feat.stride(2)
            and feat.data_ptr() % 16 == 0):
        fld = feat.stride(2)                       # channel slice of a wider map: no copy
    else:
        feat = _dev(feat, "feat", None)
    rois = _dev(rois, "rois")
    ch_scale = _dev(ch_scale, "ch_scale") if ch_scale is not None else None
    ch_shift = _dev(ch_shift, "ch_shift") if ch_shift is not None else None
    R = rois.shape[0]
    o = (output_size + bin_stride - 1) // bin_stride
    ld = C
    if out is None:
        out = torch.empty((o, o, R, C) if pos_major else (R, o, o, C), dtype=out_dtype, device=feat.device)
    else:
        if (out.dim() != 2 or tuple(out.shape) != (o * o * R, C) or out.dtype != out_dtype or not out.is_cuda
                or (R and (out.stride(1) != 1 or out.stride(0) < C or out.stride(0) % 4 or out.data_ptr() % 16))):
            raise ValueError("roi_align_nhwc: out must be a [o*o*R, C] device matrix with unit column stride")
        ld = out.stride(0) if R else C
    with torch.cuda.device(feat.device):
        check(_lib.load().locov_roi_align_nhwc_affine_fwd(_ptr(feat), _dtype_code(feat.dtype), N, H, W, C, fld, _ptr(rois), R,
                                                          output_size, output_size, float(spatial_scale),
                                                          int(sampling_ratio), int(aligned), int(bin_stride),
                                                          int(pos_major), _ptr(ch_scale), _ptr(ch_shift), int(relu),
                                                          _ptr(out), ld, _dtype_code(out_dtype), _stream(feat)),
              "locov_roi_align_nhwc_affine_fwd")
    return out


def _layout_dims(x: torch.Tensor, channels_last) -> Tuple[int, int, int, int]:
    """(R, C, HW, layout code) of a region-feature tensor: False/0 = [R,C,h,w]; True/1 = [R,h,w,C];
    2 = position-major [h,w,R,C]."""
    code = int(channels_last)
    if x.dim() == 2:
        return x.shape[0], x.shape[1], 1, 0
    if code == 2:
        R, C = x.shape[-2], x.shape[-1]
        return R, C, (x.numel() // (R * C) if R else 1), 2
    R = x.shape[0]
    C = x.shape[-1] if code == 1 else x.shape[1]
    return R, C, (x[0].numel() // C if R else 1), code


def spatial_mean(x: torch.Tensor, channels_last=False) -> torch.Tensor:
    """[R,C,h,w] (channels_last=1: [R,h,w,C]; =2: position-major [h,w,R,C]) -> [R,C]."""
    x = _dev(x, "x")
    if x.dim() == 2:
        return x
    R, C, hw, code = _layout_dims(x, channels_last)
    out = torch.empty((R, C), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.load().locov_spatial_mean_fwd(_ptr(x), R, C, max(hw, 1), code, _ptr(out), _stream(x)),
              "locov_spatial_mean_fwd")
    return out


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
           scale: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           relu: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = epi(x . weight^T): fp32 on the f32 MFMA pipe.  x [M,K] (rows may be strided), weight [N,K]."""
    x = _rows(x, "x")
    weight = _dev(weight, "weight")
    M, K = x.shape
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError(f"shape mismatch: x {tuple(x.shape)} weight {tuple(weight.shape)}")
    if K % 4 != 0:
        # the kernel stages 16-byte chunks: zero-pad the contraction dim (rows stay 16-byte aligned)
        pad = 4 - K % 4
        x = torch.nn.functional.pad(x, (0, pad)).contiguous()
        weight = torch.nn.functional.pad(weight, (0, pad))
        K += pad
    bias = _dev(bias, "bias") if bias is not None else None
    scale = _dev(scale, "scale") if scale is not None else None
    residual = _dev(residual, "residual") if residual is not None else None
    if residual is not None and tuple(residual.shape) != (M, N):
        raise ValueError("residual must be [M,N]")
    y = _out(out, (M, N), x, "linear")
    with torch.cuda.device(x.device):
        check(_lib.load().locov_gemm_nt_f32(_ptr(x), x.stride(0) if M else K, _ptr(weight), _ptr(scale), _ptr(bias), _ptr(residual),
                                            _ptr(y), N, M, N, K, _lib.EPI_RELU if relu else 0, _stream(x)),
              "locov_gemm_nt_f32")
    return y


def conv3x3_nhwc(x: torch.Tensor, w_packed: torch.Tensor, H: int, W: int, *, scale=None, shift=None,
                 residual=None, relu: bool = False, pos_major: bool = False) -> torch.Tensor:
    """3x3 / pad 1 / stride 1 convolution over R independent HxW channels-last tiles.
    x [R*H*W, Cin], w_packed [N, 9*Cin] (pack_conv3x3_weight) -> [R*H*W, N].
    Rows are ROI-major (r*H*W + pos) or, with pos_major, position-major (pos*R + r)."""
    x = _dev(x, "x")
    w_packed = _dev(w_packed, "w_packed")
    M, Cin = x.shape
    N = w_packed.shape[0]
    if w_packed.shape[1] != 9 * Cin or M % (H * W) != 0:
        raise ValueError("conv3x3_nhwc: inconsistent shapes")
    scale = _dev(scale, "scale") if scale is not None else None
    shift = _dev(shift, "shift") if shift is not None else None
    residual = _dev(residual, "residual") if residual is not None else None
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.load().locov_conv3x3_nhwc_f32(_ptr(x), M // (H * W), H, W, Cin, int(pos_major), _ptr(w_packed), _ptr(scale),
                                                 _ptr(shift), _ptr(residual), _ptr(y), N,
                                                 _lib.EPI_RELU if relu else 0, _stream(x)), "locov_conv3x3_nhwc_f32")
    return y


def pack_conv3x3_weight(w: torch.Tensor, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """[N,Cin,3,3] -> [N, 9*Cin] with k = (ky*3+kx)*Cin + c."""
    w = _dev(w, "w")
    N, Cin = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3):
        raise ValueError("pack_conv3x3_weight expects [N,Cin,3,3]")
    out = torch.empty((N, 9 * Cin), dtype=dtype, device=w.device)
    with torch.cuda.device(w.device):
        check(_lib.load().locov_pack_conv3x3_weight(_ptr(w), N, Cin, _ptr(out), _dtype_code(dtype), _stream(w)),
              "locov_pack_conv3x3_weight")
    return out


_WINO_WS = {}      # (device, stream) -> cached transform-domain workspace (grows to the largest request);
                   # per stream, because calls enqueued on different streams may run concurrently


def winograd_pack_weight(w: torch.Tensor) -> torch.Tensor:
    """[N,Cin,3,3] -> U [121, N, Cin]: the filter in the F(4,3)|F(3,3) minimal-filtering domain."""
    w = _dev(w, "w")
    N, Cin = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3):
        raise ValueError("winograd_pack_weight expects [N,Cin,3,3]")
    out = torch.empty((121, N, Cin), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        check(_lib.load().locov_winograd_pack_weight(_ptr(w), N, Cin, _ptr(out), _stream(w)), "locov_winograd_pack_weight")
    return out


def winograd_conv3x3(x: torch.Tensor, U, *, scale=None, shift=None, relu: bool = False,
                     out: Optional[torch.Tensor] = None, v_scale: float = 0.25, roi_major: bool = False,
                     in_roi_major: bool = False, out_split_scale: Optional[float] = None,
                     range_check_scale: Optional[float] = None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3x3 / pad 1 / stride 1 convolution of 7x7 position-major tiles in the Winograd domain.
    x [49*R, Cin] (row = pos*R + r), U [121, N, Cin] (winograd_pack_weight) -> [49*R, N].
    out: optional [49*R, N] destination whose rows may be a column block of a wider matrix.
    U may be a SplitWeight (split_pack(winograd_pack_weight(w))): the 121 transform-domain GEMMs then run with
    split operands on the f16 matrix pipe, the transformed input scaled by v_scale (the input transform amplifies
    non-negative data by up to 64x, any data by up to 100x: 0.25 keeps |x| < 4094 in fp16's range, as for the 1x1s).
    roi_major: write the output rows ROI-major (row = r*49 + pos), the order linear_split_segmean reads;
    in_roi_major: x is given in that order.
    out_split_scale (split U only): write the output in the split layout of split_pack scaled by that power of two -- the
    pre-split A operand (`x_is_split`) of the split GEMM that consumes it, which then stages it by LDS DMA with no conversion.
    The returned tensor is float32-TYPED storage of that layout (same shape and size), not fp32 values.
    range_check_scale (split U, fp32 output): raise the range guard here when |range_check_scale * y| >= 65504 -- the check the
    split GEMM reading y at that operand scale would make, one launch earlier.
    workspace: a caller-owned uint8 device buffer of at least winograd_workspace_bytes(R, Cin, N) bytes instead of the cached one; with
    a split U it starts with the transformed input [121, R, Cin] in the split layout x v_scale, which winograd_wgrad(v_split=...)
    takes as it is (a training step keeps one per bottleneck)."""
    x = _dev(x, "x")
    split = U if isinstance(U, SplitWeight) else None
    U = _dev(split.data if split is not None else U, "U")
    M, Cin = x.shape
    if U.dim() != 3 or U.shape[0] != 121 or U.shape[2] != Cin or M % 49 != 0:
        raise ValueError("winograd_conv3x3: inconsistent shapes")
    N, R = U.shape[1], M // 49
    scale = _dev(scale, "scale") if scale is not None else None
    shift = _dev(shift, "shift") if shift is not None else None
    if out is None:
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    else:
        y = out
        if (y.dim() != 2 or tuple(y.shape) != (M, N) or y.dtype != torch.float32 or not y.is_cuda
                or (M and (y.stride(1) != 1 or y.stride(0) < N or y.stride(0) % 2 or y.data_ptr() % 16))):
            raise ValueError("winograd_conv3x3: out must be a [49*R, N] fp32 device matrix with unit column stride")
    ldy = y.stride(0) if M else N
    lib = _lib.load()
    need = int(lib.locov_winograd_workspace_bytes(R, Cin, N))
    if workspace is not None:
        ws = workspace
        if not (ws.is_cuda and ws.device == x.device and ws.dtype == torch.uint8 and ws.is_contiguous() and ws.numel() >= need
                and ws.data_ptr() % 16 == 0):
            raise ValueError(f"winograd_conv3x3: workspace must be a contiguous uint8 device buffer of >= {need} bytes")
    else:
        key = (x.device, torch.cuda.current_stream(x.device).cuda_stream)
        ws = _WINO_WS.get(key)
        if ws is None or ws.numel() < need:
            ws = None
            _WINO_WS.pop(key, None)
            ws = _WINO_WS[key] = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
    wflags = ((_lib.EPI_RELU if relu else 0) | (_lib.WINO_OUT_ROI_MAJOR if roi_major else 0)
              | (_lib.WINO_IN_ROI_MAJOR if in_roi_major else 0))
    with torch.cuda.device(x.device):
        if split is not None:
            if out_split_scale is not None and (N % 32 or ldy != N):
                raise ValueError("winograd_conv3x3: a split-layout output needs N % 32 == 0 and a dense destination")
            check(lib.locov_winograd_conv3x3_f32_split_ex(_ptr(x), R, Cin, _ptr(U), split.scale, float(v_scale), 0, _ptr(scale),
                                                          _ptr(shift), None, _ptr(y), ldy, N, wflags,
                                                          float(out_split_scale or (-range_check_scale if range_check_scale else 0.0)),
                                                          _ptr(ws), ws.numel(),
                                                          _ptr(_overflow_word(x)), None, _stream(x)),
                  "locov_winograd_conv3x3_f32_split_ex")
        else:
            if out_split_scale is not None:
                raise ValueError("winograd_conv3x3: out_split_scale needs a split U")
            check(lib.locov_winograd_conv3x3_f32(_ptr(x), R, Cin, _ptr(U), _ptr(scale), _ptr(shift), _ptr(y), ldy, N,
                                                 wflags, _ptr(ws), ws.numel(), _stream(x)),
                  "locov_winograd_conv3x3_f32")
    return y


def winograd_workspace_bytes(R: int, Cin: int, N: int) -> int:
    """Bytes of the workspace winograd_conv3x3 needs for R tiles (locov_winograd_workspace_bytes)."""
    return int(_lib.load().locov_winograd_workspace_bytes(int(R), int(Cin), int(N)))


def conv1x1_winograd_conv3x3(x: torch.Tensor, w1: SplitWeight, b1: Optional[torch.Tensor], U: SplitWeight, *,
                             scale1: Optional[torch.Tensor] = None, scale2: Optional[torch.Tensor] = None,
                             shift2: Optional[torch.Tensor] = None, relu: bool = True, x_scale: float = 16.0, v_scale: float = 0.25,
                             roi_major: bool = True, out_split_scale: Optional[float] = None) -> torch.Tensor:
    """A bottleneck's conv1 (1x1 + FrozenBN + ReLU) and conv2 (3x3 + FrozenBN + ReLU?) in one call, split arithmetic:
        winograd_conv3x3(linear_split(x, w1, b1, scale=scale1, relu=True, x_is_split=True), U, in_roi_major=True, ...)
    and the same bits.  x [49*R, K]: the block input in the split layout (x x_scale), ROI-major rows.  Where the launch fills the
    chip with 256x256 tiles the pixel tensor between the two convolutions is never written: conv1's epilogue applies the Winograd
    input transform (gemm_split_big.hip, MODE_WINO)."""
    x = _rows(x, "x")
    w1d, Ud = _dev(w1.data, "w1"), _dev(U.data, "U")
    M, K = x.shape
    C = w1d.shape[0]
    if w1d.shape[1] != K or K % 32 or M % 49 or Ud.dim() != 3 or Ud.shape[0] != 121 or Ud.shape[2] != C or C % 32:
        raise ValueError(f"conv1x1_winograd_conv3x3: x {tuple(x.shape)} w1 {tuple(w1d.shape)} U {tuple(Ud.shape)}")
    N, R = Ud.shape[1], M // 49
    if N % 4 or (out_split_scale is not None and N % 32):
        raise ValueError("conv1x1_winograd_conv3x3: N % 4 == 0 (N % 32 == 0 for a split-layout output)")
    b1 = _dev(b1, "b1") if b1 is not None else None
    scale1 = _dev(scale1, "scale1") if scale1 is not None else None
    scale2 = _dev(scale2, "scale2") if scale2 is not None else None
    shift2 = _dev(shift2, "shift2") if shift2 is not None else None
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    need = int(lib.locov_conv1x1_winograd_workspace_bytes_for(R, K, C, N, x.stride(0) if M else K))     # no pixel scratch for the fused form
    key = (x.device, torch.cuda.current_stream(x.device).cuda_stream)
    ws = _WINO_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = None
        _WINO_WS.pop(key, None)
        ws = _WINO_WS[key] = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
    wflags = (_lib.EPI_RELU if relu else 0) | (_lib.WINO_OUT_ROI_MAJOR if roi_major else 0) | _lib.WINO_IN_ROI_MAJOR
    with torch.cuda.device(x.device):
        check(lib.locov_conv1x1_winograd_conv3x3_f32_split(_ptr(x), x.stride(0) if M else K, K, float(x_scale), _ptr(w1d), w1.scale,
                                                           _ptr(scale1), _ptr(b1), R, C, _ptr(Ud), U.scale, float(v_scale),
                                                           _ptr(scale2), _ptr(shift2), _ptr(y), N, N, wflags,
                                                           float(out_split_scale or 0.0), _ptr(ws), ws.numel(),
                                                           _ptr(_overflow_word(x)), _stream(x)),
              "locov_conv1x1_winograd_conv3x3_f32_split")
    return y


def roi_align_winograd_conv3x3(feat: torch.Tensor, rois: torch.Tensor, output_size: int, spatial_scale: float, sampling_ratio: int,
                               aligned: bool, U: SplitWeight, *, ch_scale: Optional[torch.Tensor] = None,
                               ch_shift: Optional[torch.Tensor] = None, scale2: Optional[torch.Tensor] = None,
                               shift2: Optional[torch.Tensor] = None, relu: bool = True, v_scale: float = 0.25,
                               roi_major: bool = True, out_split_scale: Optional[float] = None) -> torch.Tensor:
    """Block 0's pooler + FrozenBN + ReLU + conv2 in one call, split arithmetic:
        winograd_conv3x3(roi_align_nhwc(feat, rois, 14, ..., bin_stride=2, ch_scale, ch_shift, relu=True).view(49 R, C), U,
                         in_roi_major=True, ...)
    and the same bits.  feat [N,H,W,C] fp32, possibly a channel slice of a wider channels-last map.  With C % 64 == 0 the pooled
    rows never leave the ROIAlign workgroup: it writes their Winograd input transform itself (roi_align_nhwc.hip, WINO)."""
    if not isinstance(feat, torch.Tensor) or feat.dim() != 4 or feat.dtype != torch.float32:
        raise ValueError("roi_align_winograd_conv3x3: feat must be fp32 [N,H,W,C]")
    Nimg, H, W, C = feat.shape
    fld = C
    if (feat.is_cuda and feat.stride(3) == 1 and feat.stride(2) > C and feat.stride(2) % 4 == 0
            and feat.stride(1) == W * feat.stride(2) and feat.stride(0) == H * W * feat.stride(2)
            and feat.data_ptr() % 16 == 0):
        fld = feat.stride(2)                       # channel slice of a wider map: no copy
    else:
        feat = _dev(feat, "feat")
    rois = _dev(rois, "rois")
    Ud = _dev(U.data, "U")
    if Ud.dim() != 3 or Ud.shape[0] != 121 or Ud.shape[2] != C or C % 32 or output_size not in (13, 14):
        raise ValueError(f"roi_align_winograd_conv3x3: feat {tuple(feat.shape)} U {tuple(Ud.shape)} output_size {output_size}")
    N, R = Ud.shape[1], rois.shape[0]
    if N % 4 or (out_split_scale is not None and N % 32):
        raise ValueError("roi_align_winograd_conv3x3: N % 4 == 0 (N % 32 == 0 for a split-layout output)")
    ch_scale = _dev(ch_scale, "ch_scale") if ch_scale is not None else None
    ch_shift = _dev(ch_shift, "ch_shift") if ch_shift is not None else None
    scale2 = _dev(scale2, "scale2") if scale2 is not None else None
    shift2 = _dev(shift2, "shift2") if shift2 is not None else None
    y = torch.empty((49 * R, N), dtype=torch.float32, device=feat.device)
    lib = _lib.load()
    need = int(lib.locov_roi_align_winograd_workspace_bytes(R, C, N))       # no pixel scratch for the fused form
    key = (feat.device, torch.cuda.current_stream(feat.device).cuda_stream)
    ws = _WINO_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = None
        _WINO_WS.pop(key, None)
        ws = _WINO_WS[key] = torch.empty(max(need, 16), dtype=torch.uint8, device=feat.device)
    wflags = (_lib.EPI_RELU if relu else 0) | (_lib.WINO_OUT_ROI_MAJOR if roi_major else 0) | _lib.WINO_IN_ROI_MAJOR
    with torch.cuda.device(feat.device):
        check(lib.locov_roi_align_winograd_conv3x3_f32_split(_ptr(feat), Nimg, H, W, C, fld, _ptr(rois), R, int(output_size),
                                                             float(spatial_scale), int(sampling_ratio), int(aligned), _ptr(ch_scale),
                                                             _ptr(ch_shift), _ptr(Ud), U.scale, float(v_scale), _ptr(scale2), _ptr(shift2),
                                                             _ptr(y), N, N, wflags, float(out_split_scale or 0.0), _ptr(ws), ws.numel(),
                                                             _ptr(_overflow_word(feat)), _stream(feat)),
              "locov_roi_align_winograd_conv3x3_f32_split")
    return y


def gemm_nt_batched(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """x [B,M,K], w [B,N,K] -> [B,M,N] (fp32, one launch)."""
    x = _dev(x, "x")
    w = _dev(w, "w")
    B, M, K = x.shape
    if w.shape[0] != B or w.shape[2] != K:
        raise ValueError("gemm_nt_batched: inconsistent shapes")
    N = w.shape[1]
    y = torch.empty((B, M, N), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.load().locov_gemm_nt_batched_f32(_ptr(x), K, M * K, _ptr(w), N * K, _ptr(y), N, M * N, M, N, K, B,
                                                    _stream(x)), "locov_gemm_nt_batched_f32")
    return y


class RangeGuard:
    """One range-guard word of the split arithmetic: a device int32 every split-operand launch made while the guard is
    ACTIVE (`with ops.range_guard(g):`) ORs 1 into when it saw |scale * x| >= 65504, plus a pinned host mirror.

        g.reset()            zero the word (enqueued on the current stream)
        g.snapshot()         enqueue the 4-byte copy to pinned memory + an event (no host wait)
        g.raised()           the mirrored value; waits for the snapshot's event only (takes the snapshot first if none is pending)

    A guard belongs to ONE caller (a module and a stream): nothing else clears or reads it, so a reset enqueued by another
    module, stream or thread cannot hide a flag raised for this caller."""

    def __init__(self, device, deferred: bool = False):
        self.device = torch.device(device)
        self.word = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.event = None
        # deferred: nobody reads this word before the results are used -- whoever holds the guard zero-fills what the guarded
        # launches produced ON THE DEVICE when it is set (zero_if_raised) and reads the word whenever it next talks to the host
        # (the training forward: res5_train.Res5BlockFn / EmbeddingProposalsRes5ROIHeads.forward)
        self.deferred = deferred

    def reset(self) -> None:
        self.word.zero_()
        self.event = None

    def snapshot(self) -> None:
        self.host.copy_(self.word, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(self.device))

    def raised(self) -> bool:
        if self.event is None:
            self.snapshot()
        self.event.synchronize()
        self.event = None
        return bool(int(self.host[0]))


_GUARDS = threading.local()
_OVERFLOW = {}


class range_guard:
    """Context manager: split-operand launches on `guard.device` made inside the block raise `guard`'s word."""

    def __init__(self, guard: Optional[RangeGuard]):
        self.guard = guard

    def __enter__(self):
        st = getattr(_GUARDS, "stack", None)
        if st is None:
            st = _GUARDS.stack = []
        st.append(self.guard)
        return self.guard

    def __exit__(self, *exc):
        _GUARDS.stack.pop()
        return False


def active_guard(device) -> Optional[RangeGuard]:
    """The innermost active guard of this thread for `device` (None outside every range_guard block)."""
    for g in reversed(getattr(_GUARDS, "stack", None) or ()):
        if g is not None and g.device == torch.device(device):
            return g
    return None


def _overflow_word(ref: torch.Tensor) -> torch.Tensor:
    """The device word a split-operand launch ORs its range-guard result into: the active guard's (range_guard), else a
    per-(device, stream) word that nobody reads (launches outside every guard are unguarded by the caller's choice)."""
    g = active_guard(ref.device)
    if g is not None:
        return g.word
    key = (ref.device, torch.cuda.current_stream(ref.device).cuda_stream)
    w = _OVERFLOW.get(key)
    if w is None:
        w = _OVERFLOW[key] = torch.zeros(1, dtype=torch.int32, device=ref.device)
    return w


def split_overflow_reset(device) -> None:
    """Clear the default (unguarded-launch) word of `device` and the current stream.  Kept for callers that drive single
    launches by hand (tests, tools); modules own a RangeGuard instead."""
    _overflow_word(torch.empty(0, device=device)).zero_()


def split_overflow_raised(device) -> bool:
    """True when a split-operand launch on the current stream, outside every guard, since the last reset saw
    |x_scale * x| >= 65504 (ONE host read: it waits for the launches enqueued so far)."""
    return bool(_overflow_word(torch.empty(0, device=device)).item())


class SplitWeight(NamedTuple):
    """A weight matrix in the split-operand layout (split_pack) and the power-of-two scale it was packed with."""
    data: torch.Tensor
    scale: float


def split_pack(w: torch.Tensor, scale: Optional[float] = None) -> SplitWeight:
    """fp32 [..., K] (K % 32 == 0) -> the split-operand layout of locov_split_f16x2_pack: a float32-typed tensor of
    the same shape whose bytes hold, per row and group of 8 columns, 8 fp16 hi halves then 8 fp16 lo halves of
    scale * w (scale * w = hi + lo).  scale defaults to the power of two that puts max |scale * w| in [2^12, 2^13).
    Only meaningful as the `weight` of linear_split / gemm_nt_batched_split."""
    w = _dev(w, "w")
    K = w.shape[-1]
    if K % 32:
        raise ValueError(f"split_pack: K = {K} must be a multiple of 32")
    if scale is None:
        amax = float(w.abs().max()) if w.numel() else 1.0
        scale = 2.0 ** (12 - math.floor(math.log2(amax))) if amax > 0 and math.isfinite(amax) else 1.0
        scale = min(max(scale, 2.0 ** -100), 2.0 ** 100)
    out = torch.empty_like(w)
    rows = w.numel() // K
    with torch.cuda.device(w.device):
        check(_lib.load().locov_split_f16x2_pack(_ptr(w), rows, K, K, float(scale), _ptr(out), _ptr(_overflow_word(w)),
                                                 _stream(w)), "locov_split_f16x2_pack")
    return SplitWeight(out, float(scale))


def split_scale_for(w: torch.Tensor) -> float:
    """The power of two split_pack would choose for w (max |scale * w| in [2^12, 2^13)); ONE host read of max |w|."""
    amax = float(w.detach().abs().max()) if w.numel() else 1.0
    scale = 2.0 ** (12 - math.floor(math.log2(amax))) if amax > 0 and math.isfinite(amax) else 1.0
    return min(max(scale, 2.0 ** -100), 2.0 ** 100)


def linear_split(x: torch.Tensor, weight: SplitWeight, bias: Optional[torch.Tensor] = None, *,
                 scale: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                 relu: bool = False, x_scale: float = 16.0, x_is_split: bool = False, out_split: bool = False,
                 residual_is_split: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = epi(x . W^T), fp32 in / fp32 out, products on the f16 matrix pipe with split operands (opt-in "f16x2"
    arithmetic).  x [M,K] fp32 (rows may be strided), weight = split_pack(W [N,K]); x_scale = the power of two x is
    multiplied by before the split (|x_scale * x| must stay below 65504: |x| < 4094 at the default).
    out_split: y is written in the split layout scaled by x_scale (float32-TYPED storage of that layout: the pre-split x of
    the next split GEMM, and its residual with residual_is_split); residual_is_split: the residual is such a tensor."""
    x = _rows(x, "x")
    wd = _dev(weight.data, "weight")
    M, K = x.shape
    N = wd.shape[0]
    if wd.shape[1] != K or K % 32 or N % 4:
        raise ValueError(f"linear_split: x {tuple(x.shape)} weight {tuple(wd.shape)} (K % 32 == 0, N % 4 == 0)")
    bias = _dev(bias, "bias") if bias is not None else None
    scale = _dev(scale, "scale") if scale is not None else None
    residual = _dev(residual, "residual") if residual is not None else None
    if residual is not None and tuple(residual.shape) != (M, N):
        raise ValueError("residual must be [M,N]")
    y = _out(out, (M, N), x, "linear_split")
    with torch.cuda.device(x.device):
        check(_lib.load().locov_gemm_nt_f32_split(_ptr(x), x.stride(0) if M else K, _ptr(wd), _ptr(scale), _ptr(bias),
                                                  _ptr(residual), _ptr(y), N, M, N, K,
                                                  (_lib.EPI_RELU if relu else 0) | (_lib.GEMM_A_SPLIT if x_is_split else 0)
                                                  | (_lib.EPI_OUT_SPLIT if out_split else 0)
                                                  | (_lib.EPI_RES_SPLIT if residual_is_split and residual is not None else 0),
                                                  float(x_scale), weight.scale, _ptr(_overflow_word(x)), _stream(x)),
              "locov_gemm_nt_f32_split")
    return y


def split_unpack(t: torch.Tensor, scale: float) -> torch.Tensor:
    """fp32 values of a tensor held in the split layout (split_pack / out_split): (hi + lo) / scale.  Test / debug aid."""
    K = t.shape[-1]
    h = t.contiguous().view(torch.float16).view(*t.shape[:-1], K // 8, 2, 8).float()
    return ((h[..., 0, :] + h[..., 1, :]) / scale).reshape(t.shape)


_SEGMEAN_WS = {}


def segmean_supported(M: int, N: int, K: int, seg: int, lda: Optional[int] = None, *, residual_roi_major: bool = False,
                      x_is_split: bool = False, residual_is_split: bool = False) -> bool:
    """Can linear_split_segmean launch this shape?  (The 256 x 256 tile takes any M; the 128 x 128 one needs M * N * 4 < 2^32.)"""
    flags = ((_lib.SEGMEAN_RES_ROI_MAJOR if residual_roi_major else 0) | (_lib.GEMM_A_SPLIT if x_is_split else 0)
             | (_lib.EPI_RES_SPLIT if residual_is_split else 0) | _lib.EPI_RELU)
    return bool(_lib.load().locov_gemm_segmean_supported(int(K if lda is None else lda), int(M), int(N), int(K), int(seg), flags))


def linear_split_segmean(x: torch.Tensor, weight: SplitWeight, bias: Optional[torch.Tensor], residual: torch.Tensor, seg: int, *,
                         scale: Optional[torch.Tensor] = None, relu: bool = True, x_scale: float = 16.0,
                         residual_roi_major: bool = False, x_is_split: bool = False,
                         residual_is_split: bool = False) -> torch.Tensor:
    """Res5's last 1x1 convolution fused with the spatial mean behind it:
        out[q, :] = mean_{p < seg} relu(scale * (x[q*seg + p, :] . W^T) + bias + residual[p*R + q, :]),   R = M // seg
    x [M,K] with ROI-major rows, residual [M,N] with POSITION-major rows (the previous block's output; ROI-major rows
    q*seg + p with residual_roi_major), -> [R,N].
    The [M,N] tensor is neither written nor re-read; deterministic (fixed summation order)."""
    x = _rows(x, "x")
    wd = _dev(weight.data, "weight")
    residual = _dev(residual, "residual")
    M, K = x.shape
    N = wd.shape[0]
    if wd.shape[1] != K or K % 32 or N % 4 or seg <= 0 or M % seg or tuple(residual.shape) != (M, N):
        raise ValueError(f"linear_split_segmean: x {tuple(x.shape)} weight {tuple(wd.shape)} residual {tuple(residual.shape)} seg {seg}")
    bias = _dev(bias, "bias") if bias is not None else None
    scale = _dev(scale, "scale") if scale is not None else None
    out = torch.empty((M // seg, N), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    need = int(lib.locov_gemm_segmean_workspace_bytes(M, N))
    key = (x.device, torch.cuda.current_stream(x.device).cuda_stream)
    ws = _SEGMEAN_WS.get(key)
    if ws is None or ws.numel() < need:
        _SEGMEAN_WS.pop(key, None)
        ws = _SEGMEAN_WS[key] = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.locov_gemm_nt_f32_split_segmean(_ptr(x), x.stride(0) if M else K, _ptr(wd), _ptr(scale), _ptr(bias),
                                                  _ptr(residual), _ptr(out), M, N, K, int(seg),
                                                  (_lib.EPI_RELU if relu else 0) | (_lib.SEGMEAN_RES_ROI_MAJOR if residual_roi_major else 0)
                                                  | (_lib.GEMM_A_SPLIT if x_is_split else 0) | (_lib.EPI_RES_SPLIT if residual_is_split else 0),
                                                  float(x_scale), weight.scale, _ptr(ws), ws.numel(), _ptr(_overflow_word(x)),
                                                  _stream(x)),
              "locov_gemm_nt_f32_split_segmean")
    return out


def gemm_nt_batched_split(x: torch.Tensor, w: SplitWeight, x_scale: float = 1.0) -> torch.Tensor:
    """x [B,M,K] fp32, w = split_pack(w [B,N,K]) -> [B,M,N] fp32 (one launch, split-operand arithmetic)."""
    x = _dev(x, "x")
    wd = _dev(w.data, "w")
    B, M, K = x.shape
    if wd.shape[0] != B or wd.shape[2] != K or K % 32:
        raise ValueError("gemm_nt_batched_split: inconsistent shapes")
    N = wd.shape[1]
    y = torch.empty((B, M, N), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.load().locov_gemm_nt_batched_f32_split(_ptr(x), K, M * K, _ptr(wd), N * K, _ptr(y), N, M * N, M, N, K,
                                                          B, float(x_scale), w.scale, _ptr(_overflow_word(x)), _stream(x)),
              "locov_gemm_nt_batched_f32_split")
    return y


def frozen_bn_fold(weight, bias, running_mean, running_var, eps: float = 1e-5):
    """FrozenBatchNorm2d -> per-channel (scale, shift)."""
    weight = _dev(weight, "weight")
    C = weight.numel()
    scale = torch.empty(C, dtype=torch.float32, device=weight.device)
    shift = torch.empty(C, dtype=torch.float32, device=weight.device)
    with torch.cuda.device(weight.device):
        check(_lib.load().locov_frozen_bn_fold(_ptr(weight), _ptr(_dev(bias, "bias")),
                                               _ptr(_dev(running_mean, "running_mean")),
                                               _ptr(_dev(running_var, "running_var")), float(eps), C, _ptr(scale),
                                               _ptr(shift), _stream(weight)), "locov_frozen_bn_fold")
    return scale, shift


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """torchvision.ops.nms semantics on the device: indices of the kept boxes in descending score
    order.  The sort uses torch (plumbing); the IoU bit matrix and the greedy sweep are HIP kernels
    and nothing is copied to the host except the final count."""
    boxes = _dev(boxes, "boxes")
    scores = _dev(scores, "scores")
    K = boxes.shape[0]
    if K == 0:
        return torch.zeros((0,), dtype=torch.int64, device=boxes.device)
    order = torch.argsort(scores, descending=True, stable=True)
    sorted_boxes = boxes[order].contiguous()
    lib = _lib.load()
    ws = _workspace("nms", boxes, int(lib.locov_nms_workspace_bytes(K)))        # cached: grows to the largest request
    keep = torch.empty((K,), dtype=torch.uint8, device=boxes.device)
    num = torch.empty((1,), dtype=torch.int32, device=boxes.device)
    with torch.cuda.device(boxes.device):
        check(lib.locov_nms_sorted(_ptr(sorted_boxes), K, float(iou_threshold), _ptr(ws), _ptr(keep), _ptr(num),
                                   _stream(boxes)), "locov_nms_sorted")
    return order[keep.bool()]


DETECT_MAX_IMAGES = _lib.LABEL_MAX_IMAGES
DETECT_MAX_ROWS_PER_IMAGE, DETECT_MAX_CLASSES, DETECT_MAX_TOPK = (1 << 14) - 1, (1 << 15) - 1, _lib.DETECT_MAX_CANDIDATES
_DETECT_PINNED = {}


def detect_postprocess(probs: torch.Tensor, deltas: torch.Tensor, proposal_boxes: torch.Tensor, sizes, image_shapes, weights,
                       scale_clamp: float, score_thresh: float, nms_thresh: float, topk: int):
    """The detection post-processing of a batch on the device (csrc/detect.hip): box decoding, clipping, score threshold,
    class-wise NMS, top-k -- seven launches and ONE host read (the detections per image + the flag word) instead of the torch
    chain's ~100 launches and four reads.
    probs [R, K + 1] = softmax of the logits, deltas / proposal_boxes [R, 4] (class-agnostic regression), sizes: rows per image,
    image_shapes: (height, width) per image, weights: Box2BoxTransform's.
    Returns (boxes [B, topk, 4], scores [B, topk], classes [B, topk] int64, rows [B, topk] int64, counts: list of B ints), or None
    when the kernels flagged a case they do not take (non-finite values, more than DETECT_MAX_CANDIDATES candidates in an image):
    the caller then runs the torch chain."""
    probs, deltas, proposal_boxes = _dev(probs, "probs"), _dev(deltas, "deltas"), _dev(proposal_boxes, "proposal_boxes")
    B, R, K = len(sizes), probs.shape[0], probs.shape[1] - 1
    if not (tuple(deltas.shape) == (R, 4) and tuple(proposal_boxes.shape) == (R, 4) and sum(sizes) == R and len(image_shapes) == B):
        raise ValueError("detect_postprocess: inconsistent shapes")
    if not (0 < B <= DETECT_MAX_IMAGES and 1 <= K <= DETECT_MAX_CLASSES and 1 <= topk <= DETECT_MAX_TOPK
            and max(sizes, default=0) <= DETECT_MAX_ROWS_PER_IMAGE):
        raise ValueError("detect_postprocess: outside the kernels' limits (images, classes, rows per image or top-k)")
    lib = _lib.load()
    dev = probs.device
    offs = (ctypes.c_int * (B + 1))(0, *itertools.accumulate(int(n) for n in sizes))
    hw = (ctypes.c_float * (2 * B))(*[float(v) for shape in image_shapes for v in shape[:2]])
    nbytes = int(lib.locov_detect_postprocess_workspace_bytes(max(R, 1), B))
    ws = _workspace("detect", probs, nbytes)
    out_boxes = torch.empty((B, topk, 4), dtype=torch.float32, device=dev)
    out_scores = torch.empty((B, topk), dtype=torch.float32, device=dev)
    out_classes = torch.empty((B, topk), dtype=torch.int64, device=dev)
    out_rows = torch.empty((B, topk), dtype=torch.int64, device=dev)
    counts = torch.empty((B + 1,), dtype=torch.int32, device=dev)
    wx, wy, ww, wh = (float(w) for w in weights)
    with torch.cuda.device(dev):
        check(lib.locov_detect_postprocess(_ptr(probs), probs.stride(0), K, _ptr(deltas), _ptr(proposal_boxes), offs, hw, B, wx, wy, ww, wh,
                                           float(scale_clamp), float(score_thresh), float(nms_thresh), int(topk), _ptr(ws), ws.numel(),
                                           _ptr(out_boxes), _ptr(out_scores), _ptr(out_classes), _ptr(out_rows), _ptr(counts),
                                           _stream(probs)), "locov_detect_postprocess")
    # the ONE host read: B + 1 ints to pinned memory behind an event (the stream's later work stays queued)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    host = _DETECT_PINNED.get(key)
    if host is None or host.numel() < B + 1:
        host = _DETECT_PINNED[key] = torch.empty(max(B + 1, DETECT_MAX_IMAGES + 1), dtype=torch.int32).pin_memory()
    host[:B + 1].copy_(counts, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    ev.synchronize()
    vals = host[:B + 1].tolist()
    if vals[B]:
        return None
    return out_boxes, out_scores, out_classes, out_rows, vals[:B]


# --------------------------------------------------------------------------------------
# Backward-pass building blocks of the Res5 stage (csrc/gemm_tn.hip, res5_bwd.hip, the mask epilogue of gemm_nt.hip and
# the gradient transforms of winograd.hip); composed by locov_amd/res5_train.py.
_WS = {}


def _workspace(tag: str, ref: torch.Tensor, nbytes: int) -> torch.Tensor:
    """Cached device scratch (grows to the largest request), one per (tag, device, stream)."""
    key = (tag, ref.device, torch.cuda.current_stream(ref.device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        _WS.pop(key, None)
        ws = _WS[key] = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=ref.device)
    return ws


def linear_ex(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, scale=None, residual=None,
              mask=None, relu: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """linear() with the weight rows optionally strided (a column block of a wider matrix) and an optional `mask` [M,N]:
    the finished value is kept where mask > 0 and zeroed elsewhere (ReLU backward fused into a data-gradient GEMM)."""
    x = _rows(x, "x")
    weight = _rows(weight, "weight")
    M, K = x.shape
    N = weight.shape[0]
    if weight.shape[1] != K or K % 4:
        raise ValueError(f"linear_ex: x {tuple(x.shape)} weight {tuple(weight.shape)} (K must be a multiple of 4)")
    bias = _dev(bias, "bias") if bias is not None else None
    scale = _dev(scale, "scale") if scale is not None else None
    residual = _dev(residual, "residual") if residual is not None else None
    mask = _dev(mask, "mask") if mask is not None else None
    for t, name in ((residual, "residual"), (mask, "mask")):
        if t is not None and tuple(t.shape) != (M, N):
            raise ValueError(f"linear_ex: {name} must be [M,N]")
    y = torch.empty((M, N), dtype=torch.float32, device=x.device) if out is None else out
    if tuple(y.shape) != (M, N) or not y.is_contiguous():
        raise ValueError("linear_ex: out must be a contiguous [M,N] tensor")
    with torch.cuda.device(x.device):
        check(_lib.load().locov_gemm_nt_f32_ex(_ptr(x), x.stride(0) if M else K, _ptr(weight), weight.stride(0), _ptr(scale),
                                               _ptr(bias), _ptr(residual), _ptr(mask), _ptr(y), N, M, N, K,
                                               _lib.EPI_RELU if relu else 0, _stream(x)), "locov_gemm_nt_f32_ex")
    return y


def conv3x3_nhwc_ex(x: torch.Tensor, w_packed: torch.Tensor, H: int, W: int, *, scale=None, shift=None, residual=None,
                    mask=None, relu: bool = False, pos_major: bool = False) -> torch.Tensor:
    """conv3x3_nhwc() with the optional epilogue mask of linear_ex."""
    x = _dev(x, "x")
    w_packed = _dev(w_packed, "w_packed")
    M, Cin = x.shape
    N = w_packed.shape[0]
    if w_packed.shape[1] != 9 * Cin or M % (H * W) != 0:
        raise ValueError("conv3x3_nhwc_ex: inconsistent shapes")
    scale = _dev(scale, "scale") if scale is not None else None
    shift = _dev(shift, "shift") if shift is not None else None
    residual = _dev(residual, "residual") if residual is not None else None
    mask = _dev(mask, "mask") if mask is not None else None
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.load().locov_conv3x3_nhwc_f32_ex(_ptr(x), M // (H * W), H, W, Cin, int(pos_major), _ptr(w_packed), _ptr(scale),
                                                    _ptr(shift), _ptr(residual), _ptr(mask), _ptr(y), N,
                                                    _lib.EPI_RELU if relu else 0, _stream(x)), "locov_conv3x3_nhwc_f32_ex")
    return y


def winograd_conv3x3_ex(x: torch.Tensor, U: torch.Tensor, *, scale=None, shift=None, mask=None, relu: bool = False,
                        roi_major: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """winograd_conv3x3() on the f32 MFMA with the optional output mask (rows of x, mask and the result in one order:
    ROI-major by default)."""
    x = _dev(x, "x")
    U = _dev(U, "U")
    M, Cin = x.shape
    if U.dim() != 3 or U.shape[0] != 121 or U.shape[2] != Cin or M % 49 != 0:
        raise ValueError("winograd_conv3x3_ex: inconsistent shapes")
    N, R = U.shape[1], M // 49
    scale = _dev(scale, "scale") if scale is not None else None
    shift = _dev(shift, "shift") if shift is not None else None
    mask = _dev(mask, "mask") if mask is not None else None
    if mask is not None and tuple(mask.shape) != (M, N):
        raise ValueError("winograd_conv3x3_ex: mask must be [49*R, N]")
    y = _out(out, (M, N), x, "winograd_conv3x3_ex")
    lib = _lib.load()
    ws = _workspace("wino", x, int(lib.locov_winograd_workspace_bytes(R, Cin, N)))
    flags = (_lib.EPI_RELU if relu else 0) | ((_lib.WINO_OUT_ROI_MAJOR | _lib.WINO_IN_ROI_MAJOR) if roi_major else 0)
    with torch.cuda.device(x.device):
        check(lib.locov_winograd_conv3x3_f32_ex(_ptr(x), R, Cin, _ptr(U), _ptr(scale), _ptr(shift), _ptr(mask), _ptr(y), N, N,
                                                flags, _ptr(ws), ws.numel(), _stream(x)), "locov_winograd_conv3x3_f32_ex")
    return y


def gemm_tn(a: torch.Tensor, b: torch.Tensor, row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[N,K] = row_scale[n] * sum_m a[m,n] * b[m,k]  (a [M,N], b [M,K], rows may be strided): the weight gradient
    of a 1x1 convolution, dW = s * g^T x.  Deterministic (fixed-order reduction of the M chunks)."""
    a, b = _rows(a, "a"), _rows(b, "b")
    M, N = a.shape
    K = b.shape[1]
    if b.shape[0] != M or N % 4 or K % 4:
        raise ValueError(f"gemm_tn: a {tuple(a.shape)} b {tuple(b.shape)} (N, K must be multiples of 4)")
    row_scale = _dev(row_scale, "row_scale") if row_scale is not None else None
    out = torch.empty((N, K), dtype=torch.float32, device=a.device)
    lib = _lib.load()
    ws = _workspace("tn", a, int(lib.locov_gemm_tn_workspace_bytes(M, N, K, 1)))
    with torch.cuda.device(a.device):
        check(lib.locov_gemm_tn_f32(_ptr(a), a.stride(0) if M else N, 0, _ptr(b), b.stride(0) if M else K, 0, _ptr(out), K, 0,
                                    M, N, K, 1, _ptr(row_scale), _ptr(ws), ws.numel(), _stream(a)), "locov_gemm_tn_f32")
    return out


def winograd_wgrad(x: torch.Tensor, g: torch.Tensor, row_scale: Optional[torch.Tensor] = None,
                   roi_major: bool = True, split: bool = False, v_split: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dw [N,Cin,3,3] = row_scale[n] * d/dw of conv3x3(x) . g over R 7x7 tiles, in the Winograd domain.
    x [49*R, Cin], g [49*R, N], both in the same row order.  split: the 121 TN GEMMs in split-operand arithmetic.
    v_split (split only): the workspace winograd_conv3x3(x, SplitWeight, workspace=...) was given in the forward -- it starts with
    the transformed x in the split layout, which is then not computed again (same bits)."""
    x, g = _dev(x, "x"), _dev(g, "g")
    M, Cin = x.shape
    N = g.shape[1]
    if g.shape[0] != M or M % 49 or Cin % 4 or N % 4:
        raise ValueError("winograd_wgrad: inconsistent shapes")
    R = M // 49
    if v_split is not None and not (split and R > 0 and Cin % 8 == 0 and v_split.is_cuda and v_split.device == x.device
                                    and v_split.numel() * v_split.element_size() >= 121 * R * Cin * 4 and v_split.data_ptr() % 16 == 0):
        raise ValueError("winograd_wgrad: v_split needs the split arithmetic, Cin % 8 == 0 and the forward's workspace of the same x")
    row_scale = _dev(row_scale, "row_scale") if row_scale is not None else None
    dw = torch.empty((N, Cin, 3, 3), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    ws = _workspace("wino_wgrad", x, int(lib.locov_winograd_wgrad_workspace_bytes(R, Cin, N)))
    with torch.cuda.device(x.device):
        if v_split is not None:
            check(lib.locov_winograd_wgrad_f32_split_v(_ptr(v_split), _ptr(g), R, Cin, N, _lib.WINO_IN_ROI_MAJOR if roi_major else 0,
                                                       _ptr(row_scale), _ptr(dw), _ptr(_overflow_word(x)), _ptr(ws), ws.numel(),
                                                       _stream(x)), "locov_winograd_wgrad_f32_split_v")
        elif split and R > 0:
            check(lib.locov_winograd_wgrad_f32_split(_ptr(x), _ptr(g), R, Cin, N, _lib.WINO_IN_ROI_MAJOR if roi_major else 0,
                                                     _ptr(row_scale), _ptr(dw), _ptr(_overflow_word(x)), _ptr(ws), ws.numel(),
                                                     _stream(x)), "locov_winograd_wgrad_f32_split")
        else:
            check(lib.locov_winograd_wgrad_f32(_ptr(x), _ptr(g), R, Cin, N, _lib.WINO_IN_ROI_MAJOR if roi_major else 0,
                                               _ptr(row_scale), _ptr(dw), _ptr(ws), ws.numel(), _stream(x)),
                  "locov_winograd_wgrad_f32")
    return dw


def split_scale_from_amax(x: torch.Tensor, target_log2: float = 13.0) -> torch.Tensor:
    """Device-side operand scale of a tensor whose range is only known on the device (a gradient): a 4-float device tensor
    {s, 1/s, bits of max|x|, -} with s the power of two that puts max |s x| in [2^(target-1), 2^target).  No host read."""
    x = _dev(x, "x")
    if x.numel() % 4:
        raise ValueError("split_scale_from_amax: numel must be a multiple of 4")
    out = _scale_slot(x)
    with torch.cuda.device(x.device):
        check(_lib.load().locov_split_scale_from_amax_zeroed(_ptr(x), x.numel(), float(target_log2), _ptr(out), _stream(x)),
              "locov_split_scale_from_amax_zeroed")
    return out


def amax_bound(tensors, muls, slot: Optional[torch.Tensor] = None) -> torch.Tensor:
    """A zeroed operand-scale slot (scale_slot) whose max word holds max_i (muls[i] * max |tensors[i]|): the range of a LARGE
    tensor about to be derived from these small ones (a broadcast / masked copy), without a pass over it.  No host read.
    slot: fold into this existing (lazy) slot instead of a fresh one."""
    ts = [(_dev(t, "tensor"), float(m)) for t, m in zip(tensors, muls) if t is not None and t.numel() > 0]
    if not ts:
        raise ValueError("amax_bound: no tensor")
    if len(ts) > _lib.AMAX_BOUND_MAX or any(t.numel() % 4 for t, _ in ts):
        raise ValueError(f"amax_bound: at most {_lib.AMAX_BOUND_MAX} tensors, numel % 4 == 0")
    if slot is None:
        slot = _scale_slot(ts[0][0], lazy=True)
    ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t, _ in ts])
    ns = (ctypes.c_int64 * len(ts))(*[t.numel() for t, _ in ts])
    ms = (ctypes.c_float * len(ts))(*[m for _, m in ts])
    with torch.cuda.device(slot.device):
        check(_lib.load().locov_amax_bound(ptrs, ns, ms, len(ts), _ptr(slot), _stream(slot)), "locov_amax_bound")
    return slot


_SCALE_SLOTS = {}


def _scale_slot(ref: torch.Tensor, lazy: bool = False) -> torch.Tensor:
    """16 bytes for a device-chosen operand scale, from a per-(device, stream) ring of 2048 zeroed slots.  The reduction kernel
    (split_scale_from_amax) leaves {s, 1/s, 0, 0} behind; a `lazy` slot (scale_slot: producers fold their max into
    word 2, consumers derive the scale while word 0 is zero) stays dirty too, so the ring is zeroed every time it wraps -- one
    fill per 2048 uses, enqueued behind every GEMM that read the old contents (same stream; a training step takes ~40 slots)."""
    key = (ref.device, torch.cuda.current_stream(ref.device).cuda_stream)
    ring = _SCALE_SLOTS.get(key)
    if ring is None:
        ring = _SCALE_SLOTS[key] = [torch.zeros(2048, 4, dtype=torch.float32, device=ref.device), 0, False]
    i = ring[1]
    if i == 0 and ring[2]:
        ring[0].zero_()
        ring[2] = False
    ring[1] = (i + 1) % ring[0].shape[0]
    # every hand-out dirties the ring: the reduction kernel leaves {s, 1/s, 0, 0} behind, and a LAZY consumer derives the scale
    # from word 2 only while word 0 is zero -- a slot that carried a reduced scale in the previous cycle must not be handed out
    # lazily with that stale s in it
    ring[2] = True
    return ring[0][i]


def scale_slot(ref: torch.Tensor) -> torch.Tensor:
    """A zeroed 16-byte operand-scale slot (device) for `amax_out=`: the kernel that writes a gradient folds max |.| into it,
    the split GEMMs that read the gradient (`x_scale_dev=` / `a_scale_dev=`) derive its power-of-two scale from that."""
    return _scale_slot(_dev(ref, "ref"), lazy=True)


def linear_split_ex(x: torch.Tensor, weight: SplitWeight, bias: Optional[torch.Tensor] = None, *, scale=None, residual=None,
                    mask=None, relu: bool = False, x_scale: float = 16.0, x_scale_dev: Optional[torch.Tensor] = None,
                    amax_out: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """linear_split() with the epilogue mask of linear_ex and, optionally, the operand scale of x taken from device
    memory (split_scale_from_amax / scale_slot): the data-gradient GEMMs of the training step in split arithmetic.
    amax_out: scale_slot() that receives max |y|."""
    x = _rows(x, "x")
    wd = _dev(weight.data, "weight")
    M, K = x.shape
    N = wd.shape[0]
    if wd.shape[1] != K or K % 32 or N % 4:
        raise ValueError(f"linear_split_ex: x {tuple(x.shape)} weight {tuple(wd.shape)} (K % 32 == 0, N % 4 == 0)")
    bias = _dev(bias, "bias") if bias is not None else None
    scale = _dev(scale, "scale") if scale is not None else None
    residual = _dev(residual, "residual") if residual is not None else None
    mask = _dev(mask, "mask") if mask is not None else None
    for t_, name in ((residual, "residual"), (mask, "mask")):
        if t_ is not None and tuple(t_.shape) != (M, N):
            raise ValueError(f"linear_split_ex: {name} must be [M,N]")
    y = _out(out, (M, N), x, "linear_split_ex")
    with torch.cuda.device(x.device):
        check(_lib.load().locov_gemm_nt_f32_split_ex(_ptr(x), x.stride(0) if M else K, _ptr(wd), _ptr(scale), _ptr(bias), _ptr(residual),
                                                     _ptr(mask), _ptr(y), N, M, N, K, _lib.EPI_RELU if relu else 0, float(x_scale),
                                                     _ptr(x_scale_dev), weight.scale, _ptr(_overflow_word(x)), _ptr(amax_out), _stream(x)),
              "locov_gemm_nt_f32_split_ex")
    return y


def gemm_tn_split(a: torch.Tensor, b: torch.Tensor, row_scale: Optional[torch.Tensor], a_scale_dev: torch.Tensor,
                  b_scale: float = 16.0, b_is_split: bool = False) -> torch.Tensor:
    """gemm_tn() in split-operand arithmetic: a (gradient) scaled by a_scale_dev[0] (split_scale_from_amax), b (activation)
    by b_scale.  b_is_split: b is float32-typed storage of the split layout at that scale (split_pack / out_split), K % 8 == 0:
    transposed on its way into LDS instead of converted -- the same bits."""
    a, b = _rows(a, "a"), _rows(b, "b")
    M, N = a.shape
    K = b.shape[1]
    if b.shape[0] != M or N % 4 or K % 4 or M == 0 or (b_is_split and (K % 8 or b.stride(0) % 8)):
        raise ValueError(f"gemm_tn_split: a {tuple(a.shape)} b {tuple(b.shape)} (N, K multiples of 4 -- 8 for a pre-split b --, M > 0)")
    row_scale = _dev(row_scale, "row_scale") if row_scale is not None else None
    out = torch.empty((N, K), dtype=torch.float32, device=a.device)
    lib = _lib.load()
    ws = _workspace("tn", a, int(lib.locov_gemm_tn_workspace_bytes(M, N, K, 1)))
    with torch.cuda.device(a.device):
        fn = lib.locov_gemm_tn_f32_split_b if b_is_split else lib.locov_gemm_tn_f32_split
        check(fn(_ptr(a), a.stride(0), 0, _ptr(b), b.stride(0), 0, _ptr(out), K, 0, M, N, K, 1, _ptr(row_scale),
                 _ptr(a_scale_dev), float(b_scale), _ptr(_overflow_word(a)), _ptr(ws), ws.numel(), _stream(a)),
              "locov_gemm_tn_f32_split_b" if b_is_split else "locov_gemm_tn_f32_split")
    return out


def winograd_conv3x3_split_ex(x: torch.Tensor, U: SplitWeight, *, scale=None, shift=None, mask=None, relu: bool = False,
                              roi_major: bool = True, v_scale: Optional[float] = None,
                              amax_out: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """winograd_conv3x3() in split arithmetic with the output mask; v_scale None = chosen on the device from max |V| (the input
    is a gradient)."""
    x = _dev(x, "x")
    Ud = _dev(U.data, "U")
    M, Cin = x.shape
    if Ud.dim() != 3 or Ud.shape[0] != 121 or Ud.shape[2] != Cin or M % 49 != 0:
        raise ValueError("winograd_conv3x3_split_ex: inconsistent shapes")
    N, R = Ud.shape[1], M // 49
    scale = _dev(scale, "scale") if scale is not None else None
    shift = _dev(shift, "shift") if shift is not None else None
    mask = _dev(mask, "mask") if mask is not None else None
    if mask is not None and tuple(mask.shape) != (M, N):
        raise ValueError("winograd_conv3x3_split_ex: mask must be [49*R, N]")
    y = _out(out, (M, N), x, "winograd_conv3x3_split_ex")
    lib = _lib.load()
    ws = _workspace("wino", x, int(lib.locov_winograd_workspace_bytes(R, Cin, N)))
    flags = (_lib.EPI_RELU if relu else 0) | ((_lib.WINO_OUT_ROI_MAJOR | _lib.WINO_IN_ROI_MAJOR) if roi_major else 0)
    with torch.cuda.device(x.device):
        check(lib.locov_winograd_conv3x3_f32_split_ex(_ptr(x), R, Cin, _ptr(Ud), U.scale, float(v_scale or 1.0), int(v_scale is None),
                                                      _ptr(scale), _ptr(shift), _ptr(mask), _ptr(y), N, N, flags, 0.0, _ptr(ws), ws.numel(),
                                                      _ptr(_overflow_word(x)), _ptr(amax_out), _stream(x)), "locov_winograd_conv3x3_f32_split_ex")
    return y


PREP_KINDS = {"plain": _lib.PREP_PLAIN, "t": _lib.PREP_TRANSPOSE, "col": _lib.PREP_IM2COL, "flip9": _lib.PREP_IM2COL_FLIP,
              "wino": _lib.PREP_WINO, "uflip": _lib.PREP_WINO_FLIP}


def prep_shape(kind: str, w: torch.Tensor) -> Tuple[int, ...]:
    """Shape of the operand locov_res5_weight_prep derives from the convolution weight w for `kind`."""
    N, C = w.shape[0], w.shape[1]
    return {"plain": (N, C), "t": (C, N), "col": (N, 9 * C), "flip9": (C, 9 * N), "wino": (121, N, C), "uflip": (121, C, N)}[kind]


def res5_weight_prep(jobs) -> None:
    """jobs: [(kind, weight [N,K] | [N,Cin,3,3], FrozenBN row scale | None, out, split scale)] -- every split-layout operand of a
    training step in ONE launch (locov_res5_weight_prep); `out` are float32-typed tensors of prep_shape(kind, weight)."""
    if not jobs:
        return
    lib = _lib.load()
    ref = jobs[0][1]
    for i in range(0, len(jobs), _lib.WEIGHT_PREP_MAX_JOBS):
        chunk = jobs[i:i + _lib.WEIGHT_PREP_MAX_JOBS]
        arr = (_lib.WeightPrepJob * len(chunk))()
        for a, (kind, w, rs, out, scale) in zip(arr, chunk):
            w = _dev(w, "weight")
            if tuple(out.shape) != prep_shape(kind, w) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != w.device:
                raise ValueError(f"res5_weight_prep: out of a {kind!r} job must be a contiguous fp32 {prep_shape(kind, w)} tensor")
            a.w, a.row_scale, a.out = w.data_ptr(), (_dev(rs, "row_scale").data_ptr() if rs is not None else None), out.data_ptr()
            a.scale, a.kind, a.N, a.K = float(scale), PREP_KINDS[kind], int(w.shape[0]), int(w.shape[1])
        with torch.cuda.device(ref.device):
            check(lib.locov_res5_weight_prep(arr, len(chunk), _ptr(_overflow_word(ref)), _stream(ref)), "locov_res5_weight_prep")


def weight_transpose_scale(w: torch.Tensor, row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """w [N,K] -> [K,N] with out[k,n] = row_scale[n] * w[n,k]."""
    w = _dev(w, "w")
    N, K = w.shape
    row_scale = _dev(row_scale, "row_scale") if row_scale is not None else None
    out = torch.empty((K, N), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        check(_lib.load().locov_weight_transpose_scale(_ptr(w), N, K, _ptr(row_scale), _ptr(out), _stream(w)),
              "locov_weight_transpose_scale")
    return out


def conv3x3_weight_flip(w: torch.Tensor, row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """w [N,Cin,3,3] -> [Cin,N,3,3] with out[c,n,a,b] = row_scale[n] * w[n,c,2-a,2-b]: the filter of the data gradient."""
    w = _dev(w, "w")
    N, Cin = w.shape[:2]
    row_scale = _dev(row_scale, "row_scale") if row_scale is not None else None
    out = torch.empty((Cin, N, 3, 3), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        check(_lib.load().locov_conv3x3_weight_flip(_ptr(w), N, Cin, _ptr(row_scale), _ptr(out), _stream(w)),
              "locov_conv3x3_weight_flip")
    return out


def im2col3x3(x: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """x [R*H*W, C] ROI-major pixel rows -> [R*H*W, 9*C] 3x3 / pad 1 patches (column = tap*C + c)."""
    x = _dev(x, "x")
    M, C = x.shape
    if M % (H * W) or C % 4:
        raise ValueError("im2col3x3: inconsistent shapes")
    col = torch.empty((M, 9 * C), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.load().locov_im2col3x3_nhwc(_ptr(x), M // (H * W), H, W, C, _ptr(col), _stream(x)), "locov_im2col3x3_nhwc")
    return col


def conv3x3_wgrad_unpack(dw_packed: torch.Tensor, row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[N, 9*Cin] (column = tap*Cin + c) -> [N,Cin,3,3], rows scaled."""
    dw_packed = _dev(dw_packed, "dw_packed")
    N, K9 = dw_packed.shape
    Cin = K9 // 9
    row_scale = _dev(row_scale, "row_scale") if row_scale is not None else None
    dw = torch.empty((N, Cin, 3, 3), dtype=torch.float32, device=dw_packed.device)
    with torch.cuda.device(dw.device):
        check(_lib.load().locov_conv3x3_wgrad_unpack(_ptr(dw_packed), N, Cin, _ptr(row_scale), _ptr(dw), _stream(dw)),
              "locov_conv3x3_wgrad_unpack")
    return dw


def relu_mask(g: torch.Tensor, act: torch.Tensor, amax_out: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """g where act > 0, else 0.  amax_out: scale_slot() that receives max |result|."""
    g, act = _dev(g, "g"), _dev(act, "act")
    if g.shape != act.shape or g.numel() % 4:
        raise ValueError("relu_mask: shapes must match, numel % 4 == 0")
    out = _out(out, tuple(g.shape), g, "relu_mask")
    with torch.cuda.device(g.device):
        check(_lib.load().locov_relu_mask(_ptr(g), _ptr(act), g.numel(), _ptr(out), _ptr(amax_out), _stream(g)), "locov_relu_mask")
    return out


LABEL_MAX_IMAGES, LABEL_MAX_THRESHOLDS = _lib.LABEL_MAX_IMAGES, _lib.LABEL_MAX_THRESHOLDS
AMAX_BOUND_MAX = _lib.AMAX_BOUND_MAX


def label_proposals(boxes: torch.Tensor, n_props, gt_boxes: torch.Tensor, gt_classes: torch.Tensor, n_gt, thresholds, labels_of,
                    num_classes: int, rnd: torch.Tensor):
    """Labelling of a whole batch in one launch (locov_label_proposals; roi_emb_heads.py:25-118's device half).
    boxes [sum R, 4] / gt_boxes [sum M, 4] fp32 and gt_classes [sum M] int64, concatenated over the images; n_props / n_gt: per-image
    counts (host ints); thresholds: the Matcher's [-inf, t1, ..., inf]; labels_of: its interval labels; rnd [2, sum R] float64 in [0, 1).
    Returns (gt_index, labels, key_pos, key_neg, rows [B, 4] int64) -- see include/locov_hip.h."""
    boxes = _dev(boxes, "boxes")
    B, total = len(n_props), int(sum(n_props))
    if B > _lib.LABEL_MAX_IMAGES or len(labels_of) > _lib.LABEL_MAX_THRESHOLDS or boxes.shape[0] != total:
        raise ValueError("label_proposals: too many images / matcher intervals, or counts that do not add up")
    dev = boxes.device
    gt_boxes = _dev(gt_boxes, "gt_boxes") if sum(n_gt) else None
    gt_classes = _dev(gt_classes, "gt_classes", torch.int64) if sum(n_gt) else None
    rnd = _dev(rnd, "rnd", torch.float64)
    gt_index = torch.empty(total, dtype=torch.int64, device=dev)
    labels = torch.empty(total, dtype=torch.int64, device=dev)
    keys = torch.empty((2, total), dtype=torch.float64, device=dev)
    rows = torch.zeros((B, 4), dtype=torch.int64, device=dev)
    roff = (ctypes.c_int * (B + 1))(*([0] + list(itertools.accumulate(int(n) for n in n_props))))
    goff = (ctypes.c_int * (B + 1))(*([0] + list(itertools.accumulate(int(n) for n in n_gt))))
    nt = len(labels_of)
    lo = (ctypes.c_float * max(nt, 1))(*[float(v) for v in thresholds[:-1]])
    hi = (ctypes.c_float * max(nt, 1))(*[float(v) for v in thresholds[1:]])
    lab = (ctypes.c_int * max(nt, 1))(*[int(v) for v in labels_of])
    with torch.cuda.device(dev):
        check(_lib.load().locov_label_proposals(_ptr(boxes), roff, _ptr(gt_boxes), _ptr(gt_classes), goff, B, lo, hi, lab, nt,
                                                int(num_classes), _ptr(rnd), _ptr(gt_index), _ptr(labels), _ptr(keys[0]), _ptr(keys[1]),
                                                _ptr(rows), _stream(boxes)), "locov_label_proposals")
    return gt_index, labels, keys[0], keys[1], rows


SAMPLE_MAX_PROPOSALS = _lib.SAMPLE_MAX_PROPOSALS


def sample_proposals(key_pos: torch.Tensor, key_neg: torch.Tensor, labels: torch.Tensor, gt_index: torch.Tensor, rows: torch.Tensor,
                     boxes: torch.Tensor, gt_boxes: Optional[torch.Tensor], n_props, n_gt, budget: int, max_pos: int, num_classes: int,
                     field: Optional[torch.Tensor] = None):
    """The sampler behind label_proposals for a batch whose every image fills its budget, every field of the sampled Instances
    from one launch and no host read (locov_sample_proposals; roi_emb_heads.py:79-106).  Inputs: label_proposals' outputs and inputs
    of the same batch; field: one more fp32 per-proposal field (objectness_logits).  Returns (picked, boxes [B*budget, 4], classes,
    matched gt boxes [B*budget, 4], fg flags, rois [B*budget, 5], field[picked] | None), image-major."""
    boxes = _dev(boxes, "boxes")
    B, total = len(n_props), int(sum(n_props))
    if B > _lib.LABEL_MAX_IMAGES or boxes.shape[0] != total or not n_props or max(n_props) > SAMPLE_MAX_PROPOSALS or min(n_props) < 1:
        raise ValueError("sample_proposals: 1..64 images of 1..4096 proposals each, counts that add up")
    dev = boxes.device
    key_pos, key_neg = _dev(key_pos, "key_pos", torch.float64), _dev(key_neg, "key_neg", torch.float64)
    labels, gt_index, rows = _dev(labels, "labels", torch.int64), _dev(gt_index, "gt_index", torch.int64), _dev(rows, "rows", torch.int64)
    gt_boxes = _dev(gt_boxes, "gt_boxes") if sum(n_gt) else None
    field = _dev(field, "field") if field is not None else None
    n = B * int(budget)
    picked = torch.empty(n, dtype=torch.int64, device=dev)
    out_boxes = torch.empty((n, 4), dtype=torch.float32, device=dev)
    classes = torch.empty(n, dtype=torch.int64, device=dev)
    out_gt = torch.empty((n, 4), dtype=torch.float32, device=dev)
    fg = torch.empty(n, dtype=torch.int64, device=dev)
    rois = torch.empty((n, 5), dtype=torch.float32, device=dev)
    field_out = torch.empty(n, dtype=torch.float32, device=dev) if field is not None else None
    roff = (ctypes.c_int * (B + 1))(*([0] + list(itertools.accumulate(int(v) for v in n_props))))
    goff = (ctypes.c_int * (B + 1))(*([0] + list(itertools.accumulate(int(v) for v in n_gt))))
    with torch.cuda.device(dev):
        check(_lib.load().locov_sample_proposals(_ptr(key_pos), _ptr(key_neg), _ptr(labels), _ptr(gt_index), _ptr(rows), _ptr(boxes),
                                                 _ptr(gt_boxes), _ptr(field), roff, goff, B, int(budget), int(max_pos), int(num_classes),
                                                 _ptr(picked), _ptr(out_boxes), _ptr(classes), _ptr(out_gt), _ptr(fg), _ptr(rois),
                                                 _ptr(field_out), _stream(boxes)), "locov_sample_proposals")
    return picked, out_boxes, classes, out_gt, fg, rois, field_out


def zero_if_raised(tensors, word: torch.Tensor) -> None:
    """Zero-fill every tensor of `tensors` (contiguous fp32 device tensors, None entries skipped) ON THE DEVICE when the
    range-guard word `word` is set; a no-op launch otherwise.  No host read (locov_zero_if_raised)."""
    ts = [t for t in tensors if t is not None and t.numel() > 0]
    if not ts:
        return
    for t in ts:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise LocovError("zero_if_raised: contiguous fp32 device tensors only")
    lib = _lib.load()
    with torch.cuda.device(word.device):
        for i in range(0, len(ts), _lib.ZERO_LIST_MAX):
            chunk = ts[i:i + _lib.ZERO_LIST_MAX]
            ptrs = (ctypes.c_void_p * len(chunk))(*[t.data_ptr() for t in chunk])
            counts = (ctypes.c_int64 * len(chunk))(*[t.numel() for t in chunk])
            check(lib.locov_zero_if_raised(ptrs, counts, len(chunk), _ptr(word), _stream(word)), "locov_zero_if_raised")


def spatial_mean_bwd(g: torch.Tensor, act: Optional[torch.Tensor], hw: int, amax_out: Optional[torch.Tensor] = None,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """g [R,C] -> [R*hw, C] ROI-major rows: g[r]/hw broadcast over the positions, zeroed where act <= 0."""
    g = _dev(g, "g")
    R, C = g.shape
    act = _dev(act, "act") if act is not None else None
    if act is not None and tuple(act.shape) != (R * hw, C):
        raise ValueError("spatial_mean_bwd: act must be [R*hw, C]")
    out = _out(out, (R * hw, C), g, "spatial_mean_bwd")
    with torch.cuda.device(g.device):
        check(_lib.load().locov_spatial_mean_bwd(_ptr(g), _ptr(act), R, C, hw, _ptr(out), _ptr(amax_out), _stream(g)), "locov_spatial_mean_bwd")
    return out


def rows_stride2(src: torch.Tensor, N: int, H: int, W: int, forward: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """forward: channels-last map [N,H,W,C] -> rows of its even pixels [N*OH*OW, C]; else the adjoint (rows -> zero-filled map)."""
    src = _dev(src, "src")
    C = src.shape[-1]
    OH, OW = (H + 1) // 2, (W + 1) // 2
    out = _out(out, (N * OH * OW, C) if forward else (N, H, W, C), src, "rows_stride2")
    with torch.cuda.device(src.device):
        check(_lib.load().locov_rows_stride2(_ptr(src), N, H, W, C, int(forward), _ptr(out), _stream(src)), "locov_rows_stride2")
    return out


def roi_align_nhwc_bwd(grad_rows: torch.Tensor, feat_shape, rois: torch.Tensor, output_size: int, spatial_scale: float,
                       sampling_ratio: int = 0, aligned: bool = True, bin_stride: int = 1, pos_major: bool = False,
                       accumulate_into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Adjoint of roi_align_nhwc: grad_rows [o*o*R, C] -> gradient of the channels-last map [N,H,W,C].
    accumulate_into: an existing [N,H,W,C] gradient the result is ADDED to (fp32 atomics, as into the zero-filled fresh one)."""
    grad_rows = _rows(grad_rows, "grad_rows")
    rois = _dev(rois, "rois")
    N, H, W, C = feat_shape
    if accumulate_into is None:
        gf = torch.zeros((N, H, W, C), dtype=torch.float32, device=grad_rows.device)
    else:
        gf = _out(accumulate_into, (N, H, W, C), grad_rows, "roi_align_nhwc_bwd")
    with torch.cuda.device(gf.device):
        check(_lib.load().locov_roi_align_nhwc_bwd(_ptr(grad_rows), grad_rows.stride(0) if grad_rows.shape[0] else C, N, H, W, C,
                                                   _ptr(rois), rois.shape[0], output_size, output_size, float(spatial_scale),
                                                   int(sampling_ratio), int(aligned), int(bin_stride), int(pos_major), _ptr(gf),
                                                   _stream(gf)), "locov_roi_align_nhwc_bwd")
    return gf


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    """[N,H,W,C] -> [N,C,H,W] (the transpose kernel of nchw_to_nhwc with the roles of HW and C swapped)."""
    x = _dev(x, "x")
    N, H, W, C = x.shape
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.load().locov_nchw_to_nhwc(_ptr(x), N, H * W, C, 1, _ptr(out), F32, _stream(x)), "locov_nchw_to_nhwc")
    return out


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on the f32 MFMA NT-GEMM kernel; backward: grad_x = g W as an NT GEMM against the transposed
    weight (one small weight-sized transpose), grad_W = g^T x on the TN kernel (no activation-sized transposes),
    grad_b = sum g."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = gw = gb = None
        N, K = weight.shape
        if ctx.needs_input_grad[0]:
            gx = linear(g, weight_transpose_scale(weight.detach()))           # [M,N] . ([K,N])^T
        if ctx.needs_input_grad[1]:
            if N % 4 == 0 and K % 4 == 0:
                gw = gemm_tn(g, x)                                            # g^T x, contraction over the rows
            else:                                                             # odd sizes (e.g. an 81-way cls_score): transposed copies
                gw = linear(g.t().contiguous(), x.t().contiguous())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(dim=0)
        return gx, gw, gb


def linear_autograd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Differentiable x W^T + b (both operands may require grad)."""
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        return _LinearFn.apply(x, weight, bias)
    return linear(x.detach(), weight.detach(), bias.detach() if bias is not None else None)


class _GroundingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, S, cmask, rmask, B, T, NR, temperature):
        S = _dev(S, "S")
        cmask, rmask = _dev(cmask, "caption_mask"), _dev(rmask, "region_mask")
        w2r = torch.empty((B, B), dtype=torch.float32, device=S.device)
        r2w = torch.empty((B, B), dtype=torch.float32, device=S.device)
        with torch.cuda.device(S.device):
            check(_lib.load().locov_grounding_fwd(_ptr(S), B, T, NR, _ptr(cmask), _ptr(rmask), float(temperature),
                                                  _ptr(w2r), _ptr(r2w), _stream(S)), "locov_grounding_fwd")
        ctx.save_for_backward(S, cmask, rmask)
        ctx.dims = (B, T, NR, temperature)
        return w2r, r2w

    @staticmethod
    def backward(ctx, g_w2r, g_r2w):
        S, cmask, rmask = ctx.saved_tensors
        B, T, NR, temperature = ctx.dims
        g_w2r = _dev(g_w2r if g_w2r is not None else torch.zeros(B, B, device=S.device), "grad_w2r")
        g_r2w = _dev(g_r2w if g_r2w is not None else torch.zeros(B, B, device=S.device), "grad_r2w")
        dS = torch.empty_like(S)
        with torch.cuda.device(S.device):
            check(_lib.load().locov_grounding_bwd(_ptr(S), B, T, NR, _ptr(cmask), _ptr(rmask), float(temperature),
                                                  _ptr(g_w2r), _ptr(g_r2w), _ptr(dS), _stream(S)),
                  "locov_grounding_bwd")
        return dS, None, None, None, None, None, None


def grounding_costs(S: torch.Tensor, caption_mask: torch.Tensor, region_mask: torch.Tensor,
                    temperature: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """S [B*T, B*NR] (caption tokens x region embeddings), masks [B,T] / [B,NR] fp32 ->
    (cost_w2r, cost_r2w) [B,B] with rows = captions, columns = images.  Differentiable in S."""
    B, T = caption_mask.shape
    NR = region_mask.shape[1]
    if tuple(S.shape) != (B * T, B * NR):
        raise ValueError(f"S must be [{B * T},{B * NR}], got {tuple(S.shape)}")
    return _GroundingFn.apply(S, caption_mask.to(torch.float32), region_mask.to(torch.float32), B, T, NR,
                              float(temperature))


GROUNDING_CE_MAX_B = 64        # LOCOV_GROUNDING_CE_MAX_B


class _GroundingCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cost_w2r, cost_r2w, cmask, rmask):
        ref = cost_w2r if cost_w2r is not None else cost_r2w
        c0 = _dev(cost_w2r, "cost_w2r") if cost_w2r is not None else None
        c1 = _dev(cost_r2w, "cost_r2w") if cost_r2w is not None else None
        cmask, rmask = _dev(cmask, "caption_mask"), _dev(rmask, "region_mask")
        B, T, NR = ref.shape[0], cmask.shape[1], rmask.shape[1]
        out = torch.zeros(8, dtype=torch.float32, device=ref.device) if (c0 is None or c1 is None) else \
            torch.empty(8, dtype=torch.float32, device=ref.device)
        with torch.cuda.device(ref.device):
            check(_lib.load().locov_grounding_ce_fwd(_ptr(c0), _ptr(c1), _ptr(cmask), _ptr(rmask), B, T, NR, _ptr(out), _stream(ref)),
                  "locov_grounding_ce_fwd")
        ctx.save_for_backward(*(t for t in (c0, c1) if t is not None), cmask, rmask)
        ctx.have = (c0 is not None, c1 is not None)
        vals = out.unbind(0)
        ctx.mark_non_differentiable(vals[2], vals[3], vals[6], vals[7])
        return vals

    @staticmethod
    def backward(ctx, *g):
        saved = list(ctx.saved_tensors)
        c0 = saved.pop(0) if ctx.have[0] else None
        c1 = saved.pop(0) if ctx.have[1] else None
        cmask, rmask = saved
        ref = c0 if c0 is not None else c1
        B, T, NR = ref.shape[0], cmask.shape[1], rmask.shape[1]
        ups = [(_dev(g[i].reshape(1), "grad") if g[i] is not None else None) for i in (0, 1, 4, 5)]
        d0 = torch.empty_like(c0) if c0 is not None else None
        d1 = torch.empty_like(c1) if c1 is not None else None
        with torch.cuda.device(ref.device):
            check(_lib.load().locov_grounding_ce_bwd(_ptr(c0), _ptr(c1), _ptr(cmask), _ptr(rmask), B, T, NR, *(_ptr(u) for u in ups),
                                                     _ptr(d0), _ptr(d1), _stream(ref)), "locov_grounding_ce_bwd")
        return d0, d1, None, None


def grounding_ce(cost_w2r: Optional[torch.Tensor], cost_r2w: Optional[torch.Tensor], caption_mask: torch.Tensor,
                 region_mask: torch.Tensor) -> Tuple[torch.Tensor, ...]:
    """The cross-entropy tail of GroundingHead.forward (grounding_head.py:239-290,357-377) on the [B, B] costs of grounding_costs, one
    launch (and one in backward): returns 8 scalars -- for w2r, then r2w: CE choose caption, CE choose image, batch accuracy choose
    caption, batch accuracy choose image (zeros for a cost that is None).  Differentiable in the costs."""
    ref = cost_w2r if cost_w2r is not None else cost_r2w
    if ref is None or ref.dim() != 2 or ref.shape[0] != ref.shape[1] or ref.shape[0] > GROUNDING_CE_MAX_B:
        raise ValueError(f"grounding_ce: costs must be [B, B] with B <= {GROUNDING_CE_MAX_B}")
    return _GroundingCEFn.apply(cost_w2r, cost_r2w, caption_mask.to(torch.float32), region_mask.to(torch.float32))


class _BoxRegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, proposal_boxes, gt_boxes, gt_classes, num_classes, weights, beta):
        pred = _dev(pred, "pred_deltas")
        pb, gb = _dev(proposal_boxes.detach().float(), "proposal_boxes"), _dev(gt_boxes.detach().float(), "gt_boxes")
        cls = _dev(gt_classes, "gt_classes", torch.int64)
        R, ld = pred.shape
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        need = ctx.needs_input_grad[0]
        dpred = None
        if need:
            dpred = torch.empty_like(pred) if ld == 4 else torch.zeros_like(pred)
        with torch.cuda.device(pred.device):
            check(_lib.load().locov_box_reg_loss(_ptr(pb), _ptr(gb), _ptr(pred), ld, _ptr(cls), R, int(num_classes),
                                                 *(float(w) for w in weights), float(beta), _ptr(loss), _ptr(dpred), _stream(pred)),
                  "locov_box_reg_loss")
        ctx.dpred = dpred
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return (ctx.dpred * g if ctx.dpred is not None else None), None, None, None, None, None, None


def box_reg_loss(pred_deltas: torch.Tensor, proposal_boxes: torch.Tensor, gt_boxes: torch.Tensor, gt_classes: torch.Tensor,
                 num_classes: int, weights, smooth_l1_beta: float) -> torch.Tensor:
    """[D2-upstream] FastRCNNOutputLayers.box_reg_loss ("smooth_l1") in one launch: get_deltas of the foreground rows
    (0 <= gt_classes < num_classes), smooth-L1 against pred_deltas [R, 4] or [R, 4 * num_classes], sum / max(R, 1).  The foreground
    boxes must already be known to have positive width and height.  Differentiable in pred_deltas."""
    R = pred_deltas.shape[0]
    if pred_deltas.dim() != 2 or pred_deltas.shape[1] not in (4, 4 * num_classes) or tuple(proposal_boxes.shape) != (R, 4) \
            or tuple(gt_boxes.shape) != (R, 4) or tuple(gt_classes.shape) != (R,) or gt_classes.dtype != torch.int64:
        raise ValueError("box_reg_loss: pred_deltas [R, 4 | 4K], boxes [R, 4], gt_classes [R] int64")
    return _BoxRegLossFn.apply(pred_deltas.float(), proposal_boxes, gt_boxes, gt_classes, num_classes, tuple(weights), smooth_l1_beta)


def rownorm(x: torch.Tensor, mode: int, eps: float = 1e-12) -> torch.Tensor:
    x = _dev(x, "x")
    R, D = x.shape
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(_lib.load().locov_rownorm_fwd(_ptr(x), R, D, int(mode), float(eps), _ptr(y), _stream(x)),
              "locov_rownorm_fwd")
    return y


class _RowNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode, eps):
        x = _dev(x.detach(), "x")
        ctx.save_for_backward(x)
        ctx.args = (mode, eps)
        return rownorm(x, mode, eps)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        mode, eps = ctx.args
        g = _dev(g, "grad")
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(_lib.load().locov_rownorm_bwd(_ptr(x), _ptr(g), x.shape[0], x.shape[1], int(mode), float(eps), _ptr(dx),
                                                _stream(x)), "locov_rownorm_bwd")
        return dx, None, None


def rownorm_autograd(x: torch.Tensor, mode: int, eps: float = 1e-12) -> torch.Tensor:
    """rownorm(), differentiable in x (forward and backward on the HIP kernels)."""
    if torch.is_grad_enabled() and x.requires_grad:
        return _RowNormFn.apply(x, int(mode), float(eps))
    return rownorm(x.detach(), mode, eps)


class _PoolFcFn(torch.autograd.Function):
    """(emb, deltas) = (x W_emb^T + b_emb, x W_box^T + b_box): the predictor's two FCs on one input (box_emb_head.py:196,206)
    with their backward in ONE C call (locov_pool_fc_bwd: grad_x accumulated across both layers in the GEMM epilogue, weight
    gradients as TN GEMMs, bias gradients as column sums)."""

    @staticmethod
    def forward(ctx, x, emb_w, emb_b, box_w, box_b):
        x = _rows(x.detach(), "x").contiguous()
        ctx.save_for_backward(x, emb_w.detach(), box_w.detach())
        ctx.set_materialize_grads(False)                     # an unused output (detached class predictor) sends None, not zeros
        ctx.has_bias = (emb_b is not None, box_b is not None)
        return linear(x, emb_w.detach(), None if emb_b is None else emb_b.detach()), \
            linear(x, box_w.detach(), None if box_b is None else box_b.detach())

    @staticmethod
    def backward(ctx, g_emb, g_box):
        x, emb_w, box_w = ctx.saved_tensors
        R, C5 = x.shape
        D = emb_w.shape[0]
        need = ctx.needs_input_grad
        if g_emb is None and g_box is None:
            return None, None, None, None, None
        g_emb = _dev(g_emb, "grad_emb") if g_emb is not None else None
        g_box = _dev(g_box, "grad_deltas") if g_box is not None else None
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=x.device)
        gx = new(R, C5) if need[0] else None
        gew = new(D, C5) if need[1] and g_emb is not None else None
        geb = new(D) if need[2] and ctx.has_bias[0] and g_emb is not None else None
        gbw = new(4, C5) if need[3] and g_box is not None else None
        gbb = new(4) if need[4] and ctx.has_bias[1] and g_box is not None else None
        lib = _lib.load()
        nbytes = lib.locov_pool_fc_bwd_workspace_bytes(R, C5, D)
        ws = _workspace("pool_fc_bwd", x, nbytes)
        with torch.cuda.device(x.device):
            check(lib.locov_pool_fc_bwd(_ptr(x), R, C5, _ptr(emb_w), D, _ptr(box_w), _ptr(g_emb), _ptr(g_box), _ptr(gx), _ptr(gew),
                                        _ptr(geb), _ptr(gbw), _ptr(gbb), _ptr(ws), ws.numel(), _stream(x)), "locov_pool_fc_bwd")
        return gx, gew, geb, gbw, gbb


def pool_fc_autograd(x, emb_w, emb_b, box_w, box_b):
    """Differentiable (emb_pred(x), bbox_pred(x)); x [R,C5], C5 and D multiples of 4, bbox_pred class-agnostic ([4,C5])."""
    if box_w.shape[0] != 4 or x.shape[1] % 4 or emb_w.shape[0] % 4:
        return linear_autograd(x, emb_w, emb_b), linear_autograd(x, box_w, box_b)
    return _PoolFcFn.apply(x, emb_w, emb_b, box_w, box_b)


class _SimGemmFn(torch.autograd.Function):
    """logits = emb . bank^T (+ bias) (box_emb_head.py:211) with locov_sim_gemm_bwd as its backward."""

    @staticmethod
    def forward(ctx, emb, bank, bias):
        emb = _rows(emb.detach(), "emb").contiguous()
        ctx.save_for_backward(emb, bank.detach())
        ctx.has_bias = bias is not None
        return linear(emb, bank.detach(), None if bias is None else bias.detach())

    @staticmethod
    def backward(ctx, g):
        emb, bank = ctx.saved_tensors
        g = _dev(g, "grad_logits")
        R, D = emb.shape
        K1 = bank.shape[0]
        need = ctx.needs_input_grad
        ge = torch.empty_like(emb) if need[0] else None
        gb = torch.empty_like(bank) if need[1] and K1 % 4 == 0 else None
        lib = _lib.load()
        ws = _workspace("sim_gemm_bwd", emb, lib.locov_sim_gemm_bwd_workspace_bytes(R, D, K1))
        with torch.cuda.device(emb.device):
            check(lib.locov_sim_gemm_bwd(_ptr(g), _ptr(emb), _ptr(bank), R, D, K1, _ptr(ge), _ptr(gb), _ptr(ws), ws.numel(),
                                         _stream(emb)), "locov_sim_gemm_bwd")
        if need[1] and gb is None:                           # a trainable bank whose row count is not a multiple of 4
            gb = linear(g.t().contiguous(), emb.t().contiguous())
        return ge, gb, (g.sum(dim=0) if ctx.has_bias and need[2] else None)


def sim_gemm_autograd(emb: torch.Tensor, bank: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Differentiable fp32 similarity GEMM emb [R,D] x bank [K1,D]^T."""
    if emb.shape[1] % 4 or not (torch.is_grad_enabled() and (emb.requires_grad or bank.requires_grad)):
        return linear_autograd(emb, bank, bias)
    return _SimGemmFn.apply(emb, bank, bias)


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    x = _dev(x, "x")
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.load().locov_f32_to_bf16(_ptr(x), x.numel(), _ptr(y), _stream(x)), "locov_f32_to_bf16")
    return y


def linear_bf16(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
                scale: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                relu: bool = False) -> torch.Tensor:
    """y = epi(x . weight^T) with bf16 operands on the bf16 MFMA pipe, fp32 accumulate, fp32 epilogue and
    output (the opt-in reduced-precision form of the Res5 GEMMs).  x [M,K] bf16, weight [N,K] bf16."""
    x = _dev(x, "x", torch.bfloat16)
    weight = _dev(weight, "weight", torch.bfloat16)
    M, K = x.shape
    N = weight.shape[0]
    if weight.shape[1] != K or K % 8:
        raise ValueError(f"linear_bf16: x {tuple(x.shape)} weight {tuple(weight.shape)} (K must be a multiple of 8)")
    bias = _dev(bias, "bias") if bias is not None else None
    scale = _dev(scale, "scale") if scale is not None else None
    residual = _dev(residual, "residual") if residual is not None else None
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.load().locov_gemm_nt_bf16(_ptr(x), K, _ptr(weight), _ptr(scale), _ptr(bias), _ptr(residual), _ptr(y), N,
                                             M, N, K, _lib.EPI_RELU if relu else 0, _stream(x)), "locov_gemm_nt_bf16")
    return y


def conv3x3_nhwc_bf16(x: torch.Tensor, w_packed: torch.Tensor, H: int, W: int, *, scale=None, shift=None,
                      residual=None, relu: bool = False, pos_major: bool = False) -> torch.Tensor:
    """conv3x3_nhwc with bf16 operands (x [R*H*W, Cin] bf16, w_packed [N, 9*Cin] bf16), fp32 output."""
    x = _dev(x, "x", torch.bfloat16)
    w_packed = _dev(w_packed, "w_packed", torch.bfloat16)
    M, Cin = x.shape
    N = w_packed.shape[0]
    if w_packed.shape[1] != 9 * Cin or M % (H * W) != 0:
        raise ValueError("conv3x3_nhwc_bf16: inconsistent shapes")
    scale = _dev(scale, "scale") if scale is not None else None
    shift = _dev(shift, "shift") if shift is not None else None
    residual = _dev(residual, "residual") if residual is not None else None
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.load().locov_conv3x3_nhwc_bf16(_ptr(x), M // (H * W), H, W, Cin, int(pos_major), _ptr(w_packed),
                                                  _ptr(scale), _ptr(shift), _ptr(residual), _ptr(y), N,
                                                  _lib.EPI_RELU if relu else 0, _stream(x)), "locov_conv3x3_nhwc_bf16")
    return y


def sim_gemm_bf16(emb: torch.Tensor, bank: torch.Tensor) -> torch.Tensor:
    """logits[R,K1] = emb[R,D] . bank[K1,D]^T, bf16 operands, fp32 accumulate / output."""
    emb = _dev(emb, "emb", torch.bfloat16)
    bank = _dev(bank, "bank", torch.bfloat16)
    R, D = emb.shape
    K1 = bank.shape[0]
    if bank.shape[1] != D:
        raise ValueError("emb / bank embedding dims differ")
    out = torch.empty((R, K1), dtype=torch.float32, device=emb.device)
    with torch.cuda.device(emb.device):
        check(_lib.load().locov_sim_gemm_bf16(_ptr(emb), _ptr(bank), R, D, K1, _ptr(out), K1, _stream(emb)),
              "locov_sim_gemm_bf16")
    return out


def box_head(x: torch.Tensor, emb_w: torch.Tensor, emb_b: torch.Tensor, bbox_w: torch.Tensor,
             bbox_b: torch.Tensor, bank: torch.Tensor, bank_bf16: Optional[torch.Tensor] = None,
             norm_mode: int = NORM_NONE, sim_dtype: int = F32, channels_last=False
             ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Spatial mean + EmbeddingFastRCNNOutputLayers.forward in one C call.
    x: [R,C5] | [R,C5,h,w] | channels_last=1: [R,h,w,C5] | channels_last=2: position-major [h,w,R,C5].
    Returns (pooled [R,C5], deltas [R,4], emb [R,D], logits [R,K1])."""
    x = _dev(x, "x")
    R, C5, HW, channels_last = _layout_dims(x, channels_last)
    emb_w, emb_b = _dev(emb_w, "emb_w"), _dev(emb_b, "emb_b")
    bbox_w, bbox_b = _dev(bbox_w, "bbox_w"), _dev(bbox_b, "bbox_b")
    bank = _dev(bank, "bank")
    D, K1 = emb_w.shape[0], bank.shape[0]
    if emb_w.shape[1] != C5 or bbox_w.shape != (4, C5) or bank.shape[1] != D:
        raise ValueError("box_head: inconsistent weight shapes")
    dev = x.device
    # HW == 1: the mean is the identity and the C side skips the copy when pooled aliases x
    pooled = x.reshape(R, C5) if HW == 1 else torch.empty((R, C5), dtype=torch.float32, device=dev)
    deltas = torch.empty((R, 4), dtype=torch.float32, device=dev)
    emb = torch.empty((R, D), dtype=torch.float32, device=dev)
    logits = torch.empty((R, K1), dtype=torch.float32, device=dev)
    emb_bf16 = None
    if sim_dtype == BF16:
        if bank_bf16 is None:
            bank_bf16 = to_bf16(bank)
        bank_bf16 = _dev(bank_bf16, "bank_bf16", torch.bfloat16)
        emb_bf16 = torch.empty((R, D), dtype=torch.bfloat16, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().locov_box_head_fwd(_ptr(x), R, C5, max(HW, 1), int(channels_last), _ptr(emb_w), _ptr(emb_b),
                                             _ptr(bbox_w), _ptr(bbox_b), _ptr(bank), _ptr(bank_bf16), D, K1,
                                             int(norm_mode), int(sim_dtype), _ptr(pooled), _ptr(deltas), _ptr(emb),
                                             _ptr(emb_bf16), _ptr(logits), _stream(x)), "locov_box_head_fwd")
    return pooled, deltas, emb, logits


def token_attention(sim: torch.Tensor, tok_off: torch.Tensor, num_tok: torch.Tensor, tmax: int, temperature: float,
                    gmin: torch.Tensor, cosine: bool = False, hardmax: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Multi-token class scores of the grounding predictor (box_emb_grounding_head.py:163-225).
    sim [R,Ttot] raw token similarities; tok_off / num_tok [K1] int32; gmin: 1-element device tensor.
    Returns (scores [R,K1], attention [R,K1,tmax])."""
    sim = _dev(sim, "sim")
    tok_off, num_tok = _dev(tok_off, "tok_off", torch.int32), _dev(num_tok, "num_tok", torch.int32)
    gmin = _dev(gmin.reshape(1), "gmin")
    R, Ttot = sim.shape
    K1 = num_tok.numel()
    scores = torch.empty((R, K1), dtype=torch.float32, device=sim.device)
    att = torch.empty((R, K1, tmax), dtype=torch.float32, device=sim.device)
    with torch.cuda.device(sim.device):
        check(_lib.load().locov_token_attention_fwd(_ptr(sim), R, Ttot, _ptr(tok_off), _ptr(num_tok), K1, int(tmax),
                                                    float(temperature), int(cosine), int(hardmax), _ptr(gmin),
                                                    _ptr(scores), _ptr(att), _stream(sim)), "locov_token_attention_fwd")
    return scores, att
