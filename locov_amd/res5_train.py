"""The Res5 stage under autograd on the hand-written gfx950 kernels (forward + data gradients + weight gradients).

The LSM configuration trains the Res5 convolution weights (configs/coco_lsm.yaml:8 `FREEZE_AT: 0`; FrozenBN only
freezes the statistics) through two call sites of ovr/modeling/roi_heads/roi_emb_heads.py:
    :323      visual_grid_features = self.res5(features)            -- the whole res4 grid
    :343-344  box_features = self._shared_roi_transform(...).mean() -- the sampled proposals (4 x 200 per GPU)
The reference differentiates both through cuDNN (torch autograd).  Here every tensor is a channels-last pixel-row
matrix [R*H*W, C] (ROI-major rows) and one torch.autograd.Function covers the whole stage:

  forward   per bottleneck: 1x1 (GEMM, FrozenBN + ReLU in the epilogue) -> 3x3 (Winograd domain for 7x7 tiles, implicit
            GEMM otherwise) -> 1x1 (+ projection shortcut / identity residual, ReLU); the post-ReLU activations are kept.
  backward  with g the gradient of a block's output already masked by (output > 0):
              dW3 = s3 * g^T y2                      TN GEMM over the pixel rows (gemm_tn.hip)
              g2  = (g . s3 W3) * [y2 > 0]           NT GEMM, mask fused into the epilogue (gemm_nt.hip)
              dW2 = s2 * wgrad3x3(y1, g2)            Winograd domain (121 TN GEMMs) or im2col + TN GEMM
              g1  = conv3x3(g2, flip(s2 W2)) * [y1 > 0]
              dW1 = s1 * g1^T x ;  dWs = ss * g^T x
              gx  = (g1 . s1 W1 + g [. ss Ws]) * [x > 0]      = the masked gradient of the previous block's output
            FrozenBN scales are folded into the transposed weights (data gradients) and applied to the rows of the
            weight gradients; the statistics receive no gradient (buffers), exactly as in the reference.

Arithmetic: with `split` (RES5_DTYPE "f16x2") the forward AND the backward GEMMs form every fp32 product from (hi, lo)
f16 pairs on the f16 matrix pipe with fp32 accumulation (gemm_split.hip, gemm_tn_split.hip); a gradient tensor has no
a-priori range, so its power-of-two operand scale is chosen on the device from its max |.| (one small reduction per
gradient tensor, no host read).  Without it everything runs on the f32 MFMA.  LOCOV_RES5_BWD_F32=1 keeps the backward on
the f32 MFMA while the forward uses split operands (developer A/B).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import ops

__all__ = ["res5_rows", "res5_grid", "roi_align_even_rows", "to_nhwc", "Res5RowsFn"]


import os

_NO_WINO_BWD = bool(int(os.environ.get("LOCOV_RES5_BWD_DIRECT", "0")))      # developer A/B: 3x3 gradients in the direct form
_BWD_F32 = bool(int(os.environ.get("LOCOV_RES5_BWD_F32", "0")))              # developer A/B: backward GEMMs on the f32 MFMA


def _wino_ok(H: int, W: int, cin: int, cout: int) -> bool:
    return H == 7 and W == 7 and cin % 32 == 0 and cout % 4 == 0


class Res5RowsFn(torch.autograd.Function):
    """out = Res5(x0) on ROI-major pixel rows.  x0 [R*H*W, Cin] is the stage input already sub-sampled by block 0's
    stride (the even positions: STRIDE_IN_1X1).  Returns the rows [R*H*W, Cout], or with `pooled` their per-ROI mean [R, Cout].
    weights: the convolution weights in module order (block 0: conv1, conv2, conv3, shortcut; then conv1..conv3 of the
    following blocks) -- passed as inputs so that autograd routes their gradients."""

    @staticmethod
    def _blocks(stage, x, H, W, split):
        saved: List[torch.Tensor] = []
        cols: List[torch.Tensor] = []                 # im2col patches of y1 per block (general-grid form only)
        meta = []
        wi = 0
        for blk in stage:
            has_sc = blk.shortcut is not None
            w1, s1, b1 = stage._packed(blk.conv1)
            w3, s3, b3 = stage._packed(blk.conv3)
            c2 = blk.conv2
            y1 = stage._linear(split, x, w1, b1, scale=s1, relu=True)
            wino = _wino_ok(H, W, c2.in_channels, c2.out_channels)
            if wino:
                u2, s2, b2 = stage._packed(c2, winograd=True)
                y2 = ops.winograd_conv3x3(y1, stage._split(u2) if split else u2, scale=s2, shift=b2, relu=True,
                                          roi_major=True, in_roi_major=True)
            else:
                # general grid (the whole-grid call): 3x3 as a GEMM over im2col patches -- K = 9 Cin columns in the order of
                # the packed weight -- so that it runs in the stage's arithmetic; the patches are kept for the weight gradient
                w2, s2, b2 = stage._packed(c2)
                col = ops.im2col3x3(y1, H, W)
                y2 = stage._linear(split, col, w2, b2, scale=s2, relu=True)
            if has_sc:
                ws, ss, bs = stage._packed(blk.shortcut)
                sc = stage._linear(split, x, ws, bs, scale=ss)
            else:
                sc = x
            out = stage._linear(split, y2, w3, b3, scale=s3, residual=sc, relu=True)
            saved += [x, y1, y2, out]
            if not wino:
                cols.append(col)
            meta.append((has_sc, wino, wi))
            wi += 4 if has_sc else 3
            x = out
        return saved + cols, meta, x

    @staticmethod
    def forward(ctx, x0, stage, R, H, W, pooled, split, guard, *weights):
        """guard: None, or (on_overflow,) -- check the split arithmetic's range-guard word after the forward and repeat it on
        the f32 MFMA when an activation left fp16's range (calls on_overflow() first).  Inside a caller's own
        `ops.range_guard` block the launches raise THAT guard and nothing is read here: the caller checks once for all the
        calls it made (EmbeddingProposalsRes5ROIHeads.forward: one read for the whole-grid and the ROI call together)."""
        x = ops._dev(x0.detach(), "x0")
        own = None
        if split and guard is not None and ops.active_guard(x.device) is None:
            own = stage.range_guard("fwd", x.device)
            own.reset()
        with ops.range_guard(own):
            saved, meta, out = Res5RowsFn._blocks(stage, x, H, W, split)
        if own is not None and own.raised():
            if guard[0] is not None:
                guard[0]()
            del saved, out
            saved, meta, out = Res5RowsFn._blocks(stage, x, H, W, False)
        ctx.stage, ctx.meta, ctx.geom, ctx.pooled = stage, meta, (R, H, W), pooled
        ctx.split = bool(split) and not _BWD_F32
        # a DEFERRED guard held by the caller (the ROI heads' training forward): nothing reads it before the backward runs, so
        # the backward must not turn the inf / NaN activations of an out-of-range forward into gradients -- it zero-fills them
        # on the device when that word is set (the caller does the same with this forward's outputs)
        # (its per-forward copy `step_word`, which the caller fills at the end of ITS forward: a later forward's labelling read
        # may clear the guard's own word before this backward runs)
        active = ops.active_guard(x.device) if split else None
        ctx.skip_words = [getattr(active, "step_word", active.word)] if active is not None and getattr(active, "deferred", False) else []
        ctx.nw = len(weights)
        ctx.save_for_backward(*saved)
        if pooled:
            return ops.spatial_mean(out.view(R, H, W, out.shape[1]), channels_last=1)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if not ctx.split:
            grads = Res5RowsFn._backward(ctx, grad_out, False)
            for w in ctx.skip_words:
                ops.zero_if_raised(grads, w)
            return grads
        # Split arithmetic.  What can leave fp16's range here is not data: the activations passed the forward's guard (the
        # weight-gradient GEMMs and the Winograd transforms see the same tensors at the same scales) and every gradient's
        # operand scale is chosen on the device from its own max |g|.  Only a REMEMBERED weight scale (Res5Stage._split: chosen
        # afresh every 64 packings, 8x headroom) that stopped covering a weight could trip the guard -- the pack kernel
        # raises it.  That is recorded in the stage's "bwd" guard and nothing is read inside autograd (a host read here would
        # stall DDP's overlapped all-reduce).  Instead the pass ends with a GradScaler-style skip decided ON THE DEVICE: when the
        # word is set every gradient this pass produced (stage input and all convolution weights) is zero-filled
        # (locov_zero_if_raised: one launch that exits after a scalar load when the word is clear), so no inf / NaN can reach
        # the optimizer or the all-reduce whether or not another step follows.  The ROI heads look at the word together with
        # the ONE host read of the next step's labelling (SampleAllROIHeads.label_and_sample_proposals), warn, and drop the
        # remembered scales; `stage.backward_guard_raised()` reads it on demand (e.g. next to a trainer's loss logging).
        guard = ctx.stage.range_guard("bwd", grad_out.device)
        with ops.range_guard(guard):
            grads = Res5RowsFn._backward(ctx, grad_out, True)
        ops.zero_if_raised(grads, guard.word)          # (every gradient here is a freshly written contiguous tensor or a view of one)
        for w in ctx.skip_words:
            ops.zero_if_raised(grads, w)
        return grads

    @staticmethod
    def _backward(ctx, grad_out, sp):
        stage, (R, H, W) = ctx.stage, ctx.geom
        saved = ctx.saved_tensors
        need_x = ctx.needs_input_grad[0]
        need_w = ctx.needs_input_grad[8:]
        gw: List[Optional[torch.Tensor]] = [None] * ctx.nw
        grad_out = ops._dev(grad_out, "grad_out")
        out_last = saved[4 * len(stage) - 1]
        # Operand scales of the gradients (split arithmetic): every kernel that WRITES a gradient folds max |.| into a zeroed
        # 16-byte slot (ops.scale_slot) on its way out, and the GEMMs that read the gradient derive its power-of-two scale from
        # the slot -- no separate pass over the tensor, no host read.
        slot = (lambda ref: ops.scale_slot(ref)) if sp else (lambda ref: None)
        # gradient of the last block's output, masked by its ReLU
        # (the two element-wise kernels at the head of the chain keep the separate reduction: their waves all finish together,
        # so every one of them would issue its atomic -- measured +110 us on the grid's relu_mask against a 10 us reduction)
        g = ops.spatial_mean_bwd(grad_out, out_last, H * W) if ctx.pooled else ops.relu_mask(grad_out, out_last)
        sg = ops.split_scale_from_amax(g) if sp else None

        def keyed(t, conv, tag):                       # remembered operand scale of a per-step weight packing (Res5Stage._split)
            t._locov_key = (id(conv), tag)
            return t

        def wgrad_1x1(g_, sg_, x_, s_):                # dW = s * g^T x
            if sp and g_.shape[1] % 4 == 0 and x_.shape[1] % 4 == 0 and g_.shape[0] > 0:
                return ops.gemm_tn_split(g_, x_, s_, sg_, 16.0)
            return ops.gemm_tn(g_, x_, s_)

        def dgrad_1x1(g_, sg_, wt, conv, amax_out=None, **kw):        # (g . wt^T [+ residual]) [mask]; amax_out: slot of the result
            if sp and wt.shape[1] % 32 == 0 and wt.shape[0] % 4 == 0:
                return ops.linear_split_ex(g_, stage._split(keyed(wt, conv, "t")), x_scale_dev=sg_, amax_out=amax_out, **kw), amax_out
            y_ = ops.linear_ex(g_, wt, **kw)
            return y_, (ops.split_scale_from_amax(y_) if sp and amax_out is not None else None)

        for bi in range(len(stage) - 1, -1, -1):
            blk = stage[bi]
            has_sc, wino, wi = ctx.meta[bi]
            x, y1, y2, _ = saved[4 * bi: 4 * bi + 4]
            col = None if wino else saved[4 * len(stage) + bi]
            w1, s1, _ = stage._packed(blk.conv1)
            w3, s3, _ = stage._packed(blk.conv3)
            c2 = blk.conv2
            # conv3: dW3 = s3 * g^T y2 ; g2 = (g . s3 W3) [y2 > 0]
            if need_w[wi + 2]:
                gw[wi + 2] = wgrad_1x1(g, sg, y2, s3).view_as(blk.conv3.weight)
            g2, sg2 = dgrad_1x1(g, sg, stage._derived(blk.conv3, "wt", lambda: ops.weight_transpose_scale(w3, s3)), blk.conv3,
                                amax_out=slot(g), mask=y2)
            # conv2 (3x3): dW2 = s2 * wgrad(y1, g2) ; g1 = conv3x3(g2, flip(s2 W2)) [y1 > 0]
            _, s2, _ = stage._packed(c2)
            w2 = c2.weight.detach()
            if need_w[wi + 1]:
                if wino and c2.out_channels % 4 == 0 and not _NO_WINO_BWD:
                    gw[wi + 1] = ops.winograd_wgrad(y1, g2, s2, roi_major=True, split=sp)
                else:
                    colw = col if col is not None else ops.im2col3x3(y1, H, W)
                    if sp:
                        gw[wi + 1] = ops.conv3x3_wgrad_unpack(ops.gemm_tn_split(g2, colw, None, sg2, 16.0), s2)
                    else:
                        gw[wi + 1] = ops.conv3x3_wgrad_unpack(ops.gemm_tn(g2, colw), s2)
            wflip = stage._derived(c2, "flip", lambda: ops.conv3x3_weight_flip(w2, s2))     # [Cin, Cout, 3, 3]
            sg1 = slot(g2)
            if _wino_ok(H, W, c2.out_channels, c2.in_channels) and not _NO_WINO_BWD:
                uflip = stage._derived(c2, "uflip", lambda: ops.winograd_pack_weight(wflip))
                if sp:
                    g1 = ops.winograd_conv3x3_split_ex(g2, stage._split(keyed(uflip, c2, "uflip")), mask=y1, roi_major=True, amax_out=sg1)
                else:
                    g1 = ops.winograd_conv3x3_ex(g2, uflip, mask=y1, roi_major=True)
            else:
                # data gradient on the general grid: the same im2col GEMM with the flipped filter (im2col only copies and
                # zero-pads: the patches have g2's range)
                g1, sg1 = dgrad_1x1(ops.im2col3x3(g2, H, W), sg2, stage._derived(c2, "flip9", lambda: ops.pack_conv3x3_weight(wflip)), c2,
                                    amax_out=sg1, mask=y1)
            del g2
            # conv1 (+ shortcut): dW1 = s1 * g1^T x ; gx = (g1 . s1 W1 + shortcut path) [x > 0]
            if need_w[wi]:
                gw[wi] = wgrad_1x1(g1, sg1, x, s1).view_as(blk.conv1.weight)
            if has_sc:
                ws, ss, _ = stage._packed(blk.shortcut)
                if need_w[wi + 3]:
                    gw[wi + 3] = wgrad_1x1(g, sg, x, ss).view_as(blk.shortcut.weight)
            first = bi == 0
            if first and not need_x:
                g = None
                break
            # the input of block 0 is the pooler output (no ReLU in front of it); every other block's input is the
            # post-ReLU output of its predecessor, whose mask turns gx into that block's masked output gradient
            mask = None if first else x
            # (the slot of gx is only needed when a block below reads it: not for the stage input's gradient)
            gx, sgx = dgrad_1x1(g1, sg1, stage._derived(blk.conv1, "wt", lambda: ops.weight_transpose_scale(w1, s1)), blk.conv1,
                                amax_out=None if (first or has_sc) else slot(g1),
                                residual=None if has_sc else g, mask=None if has_sc else mask)
            if has_sc:
                gx, sgx = dgrad_1x1(g, sg, stage._derived(blk.shortcut, "wt", lambda: ops.weight_transpose_scale(ws, ss)), blk.shortcut,
                                    amax_out=None if first else slot(g),
                                    residual=gx, mask=mask)
            del g1
            g, sg = gx, sgx
        return (g, None, None, None, None, None, None, None, *gw)


def _stage_weights(stage) -> Tuple[torch.Tensor, ...]:
    ws = []
    for blk in stage:
        ws += [blk.conv1.weight, blk.conv2.weight, blk.conv3.weight]
        if blk.shortcut is not None:
            ws.append(blk.shortcut.weight)
    return tuple(ws)


def res5_rows(stage, x0: torch.Tensor, R: int, H: int, W: int, pooled: bool = False, split: bool = True,
              overflow_check: bool = True, on_overflow=None) -> torch.Tensor:
    """Differentiable Res5 on ROI-major pixel rows (see Res5RowsFn)."""
    assert stage.supports_rows_path(), "the rows path needs FrozenBN, STRIDE_IN_1X1 and ungrouped convolutions"
    guard = (on_overflow,) if overflow_check else None
    return Res5RowsFn.apply(x0, stage, R, H, W, pooled, split, guard, *_stage_weights(stage))


class _ToNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.nchw_to_nhwc(x.detach())

    @staticmethod
    def backward(ctx, g):
        return ops.nhwc_to_nchw(g)


class _ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.nhwc_to_nchw(x.detach())

    @staticmethod
    def backward(ctx, g):
        return ops.nchw_to_nhwc(g)


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[N,C,H,W] -> [N,H,W,C], differentiable."""
    return _ToNHWC.apply(x)


class _RoiAlignEvenRows(torch.autograd.Function):
    """Even-grid ROIAlign of a channels-last map -> ROI-major rows [oh*ow*R, C] (oh = P/2), differentiable in the map."""

    @staticmethod
    def forward(ctx, nhwc, rois, P, scale, sampling_ratio, aligned):
        out = ops.roi_align_nhwc(nhwc.detach(), rois, P, scale, sampling_ratio, aligned, bin_stride=2, pos_major=False)
        ctx.save_for_backward(rois)
        ctx.args = (tuple(nhwc.shape), P, scale, sampling_ratio, aligned)
        return out.view(-1, nhwc.shape[3])

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        shape, P, scale, sampling_ratio, aligned = ctx.args
        return ops.roi_align_nhwc_bwd(g, shape, rois, P, scale, sampling_ratio, aligned, bin_stride=2, pos_major=False), \
            None, None, None, None, None


def roi_align_even_rows(nhwc, rois, P, scale, sampling_ratio, aligned) -> torch.Tensor:
    return _RoiAlignEvenRows.apply(nhwc, rois, int(P), float(scale), int(sampling_ratio), bool(aligned))


class _Stride2Rows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nhwc):
        N, H, W, _ = nhwc.shape
        ctx.dims = (N, H, W)
        return ops.rows_stride2(nhwc.detach(), N, H, W, True)

    @staticmethod
    def backward(ctx, g):
        N, H, W = ctx.dims
        return ops.rows_stride2(g, N, H, W, False)


def res5_grid(stage, nhwc: torch.Tensor, split: bool = True, overflow_check: bool = True, on_overflow=None) -> torch.Tensor:
    """roi_emb_heads.py:323 -- the stage applied to the whole channels-last res4 map [N,H,W,Cin] -> logical NCHW
    [N, Cout, ceil(H/2), ceil(W/2)], differentiable in the map and the convolution weights.  Block 0's stride-2 1x1
    convolutions read the even pixels; the 3x3 convolutions run as implicit GEMMs over the (H/2 x W/2) grid."""
    N, H, W, _ = nhwc.shape
    assert stage[0].stride == 2 and stage[0].stride_in_1x1
    OH, OW = (H + 1) // 2, (W + 1) // 2
    rows = _Stride2Rows.apply(nhwc)
    y = res5_rows(stage, rows, N, OH, OW, pooled=False, split=split, overflow_check=overflow_check, on_overflow=on_overflow)
    return _ToNCHW.apply(y.view(N, OH, OW, y.shape[1]))
