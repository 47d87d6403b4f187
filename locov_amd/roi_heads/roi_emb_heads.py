"""ROI heads of the LSM path: ROIAlign, Res5, spatial mean, box predictor, losses / inference -- Detectron2's plugin surface,
gfx950 kernels underneath.

Mirrors ovr/modeling/roi_heads/roi_emb_heads.py:
    SampleAllROIHeads.label_and_sample_proposals      :25-118     -> labelling.py (re-exported here)
    EmbeddingRes5ROIHeads                              :122-306
    EmbeddingProposalsRes5ROIHeads                     :310-360
The training forward of both heads is a small retry machine around ONE host wait per step (`heads.stats` counts its legs): the
sample is speculated on the device, the forward enqueued, then validated -- a miss repeats it from the true counts, a forward
that left the f16x2 split arithmetic's range repeats it on the f32 MFMA (RES5_TRAIN_GUARD "sync"; "deferred" zero-fills instead).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
from torch import nn

from .. import ops
from ..poolers import ROIPooler, convert_boxes_to_pooler_format
from ..registry import Registry, configurable
from ..res5 import build_res5_block
from ..structures import Boxes, ShapeSpec
from .box_emb_head import build_box_predictor
from .labelling import (_EVENTS, Matcher, ROIHeads, SampleAllROIHeads, add_ground_truth_to_proposals, get_event_storage,      # noqa: F401
                        subsample_labels, subsample_order)

ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
_JOINT_FORWARD = os.environ.get("LOCOV_RES5_JOINT_FWD", "1") != "0"    # developer A/B: 0 = the two Res5 calls are forwarded one after the other

__all__ = ["ROI_HEADS_REGISTRY", "build_roi_heads", "Matcher", "subsample_labels", "add_ground_truth_to_proposals",
           "ROIHeads", "SampleAllROIHeads", "EmbeddingRes5ROIHeads", "EmbeddingProposalsRes5ROIHeads"]


@ROI_HEADS_REGISTRY.register()
class EmbeddingRes5ROIHeads(SampleAllROIHeads):
    """roi_emb_heads.py:122 -- C4 ROI heads: pooler -> res5 -> mean -> box predictor."""

    @configurable
    def __init__(self, *, in_features: List[str], pooler: ROIPooler, res5: nn.Module, box_predictor: nn.Module,
                 mask_head: Optional[nn.Module] = None, output_shape: Optional[int] = 0,
                 res5_backend: str = "hip", res5_conv3x3: str = "winograd", res5_dtype: str = "f16x2",
                 res5_overflow_check: bool = True, res5_train_guard: str = "sync", **kwargs):
        super().__init__(**kwargs)
        assert res5_backend in ("hip", "miopen") and res5_conv3x3 in ("winograd", "direct") and res5_dtype in ("fp32", "f16x2", "bf16")
        # extension: "f16x2" = fp32 GEMMs formed from split f16 operand pairs on the f16 matrix pipe (fp32-level
        # accuracy, see csrc/gemm_split.hip); "bf16" = reduced-precision GEMM operands (not a parity configuration)
        self.res5_dtype = res5_dtype
        # the split arithmetic's range guard: read the device flag once per call and, if an activation left fp16's range
        # (|x| >= 4094 at the activation scale), repeat the call on the f32 MFMA
        self.res5_overflow_check = res5_overflow_check
        # training forwards: "sync" (default) = the word is copied to pinned memory behind the Res5 calls and looked at at the END
        # of this module's forward, behind the predictor's and the losses' launches (an event wait that leaves them queued: the
        # GPU does not drain); an out-of-range forward is repeated on the f32 MFMA, i.e. every step has the reference's values.
        # "deferred" = never read inside the step: acted on on the device (outputs, this module's losses and the backward's
        # gradients are zeroed: a skipped step) and read with the next step's labelling
        mode = os.environ.get("LOCOV_RES5_TRAIN_GUARD", res5_train_guard)      # (developer A/B override, validated like the key)
        assert mode in ("deferred", "sync"), f"RES5_TRAIN_GUARD / LOCOV_RES5_TRAIN_GUARD must be 'deferred' or 'sync', got {mode!r}"
        self.res5_train_guard = mode
        self._overflow_warned = False
        self.res5_backend = res5_backend      # extension: how the Res5 convolutions run (see res5.py)
        self.res5_conv3x3 = res5_conv3x3      # extension: form of the 3x3 convolutions on the hip backend
        self.in_features = in_features
        self.pooler = pooler
        if isinstance(res5, (list, tuple)):
            res5 = nn.Sequential(*res5)
        self.res5 = res5
        self.output_shape = output_shape
        self.box_predictor = box_predictor
        self.mask_on = mask_head is not None
        if self.mask_on:
            raise NotImplementedError("mask heads are outside the LSM ROI-head path (MODEL.MASK_ON is False "
                                      "in both reference configs)")

    @classmethod
    def from_config(cls, cfg, input_shape):
        ret = ROIHeads.from_config.__func__(cls, cfg)
        in_features = ret["in_features"] = cfg.MODEL.ROI_HEADS.IN_FEATURES
        pooler_resolution = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        pooler_type = cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE
        pooler_scales = (1.0 / input_shape[in_features[0]].stride,)
        sampling_ratio = cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO
        assert not cfg.MODEL.KEYPOINT_ON
        assert len(in_features) == 1
        assert not cfg.MODEL.MASK_ON, "mask heads are outside the LSM ROI-head path"
        ret["pooler"] = ROIPooler(output_size=pooler_resolution, scales=pooler_scales,
                                  sampling_ratio=sampling_ratio, pooler_type=pooler_type)
        ret["res5"], out_channels = cls._build_res5_block(cfg)
        ret["box_predictor"] = build_box_predictor(cfg, input_shape=out_channels)
        ret["output_shape"] = out_channels
        box_head = cfg.MODEL.ROI_BOX_HEAD
        ret["res5_backend"] = box_head.get("RES5_BACKEND", "hip") if hasattr(box_head, "get") else "hip"
        ret["res5_conv3x3"] = box_head.get("RES5_CONV3X3", "winograd") if hasattr(box_head, "get") else "winograd"
        ret["res5_dtype"] = box_head.get("RES5_DTYPE", "f16x2") if hasattr(box_head, "get") else "f16x2"
        ret["res5_overflow_check"] = bool(box_head.get("RES5_OVERFLOW_CHECK", True)) if hasattr(box_head, "get") else True
        ret["res5_train_guard"] = box_head.get("RES5_TRAIN_GUARD", "sync") if hasattr(box_head, "get") else "sync"
        return ret

    @classmethod
    def _build_res5_block(cls, cfg):
        return build_res5_block(cfg)

    def _rows_path_reason(self, features: List[torch.Tensor]) -> Optional[str]:
        """None when the channels-last hand-written path applies (one feature level, FrozenBN Res5 whose block 0 strides in
        its 1x1 convs, ungrouped 3x3, even pooler resolution, channels a multiple of 32, device tensors) -- else the first
        condition that fails, in words (the text of the fallback warning)."""
        if self.res5_backend != "hip":
            return f"MODEL.ROI_BOX_HEAD.RES5_BACKEND is {self.res5_backend!r}"
        if len(features) != 1:
            return f"{len(features)} feature levels (the hand-written Res5 path takes one)"
        ph, pw = self.pooler.output_size
        if not hasattr(self.res5, "forward_rows"):
            return f"res5 is a {type(self.res5).__name__}, not locov_amd.res5.Res5Stage"
        if not self.res5.supports_rows_path():
            return "Res5 needs FrozenBN on every convolution, STRIDE_IN_1X1 and ungrouped 3x3 convolutions"
        if self.res5[0].stride != 2:
            return f"block 0 has stride {self.res5[0].stride} (the even-grid pooler assumes 2)"
        if ph != pw or ph % 2:
            return f"pooler resolution {ph}x{pw} is not square and even"
        if features[0].shape[1] % 32:
            return f"{features[0].shape[1]} input channels (a multiple of 32 is needed)"
        if not features[0].is_cuda:
            return f"the feature map is on {features[0].device}"
        return None

    def _rows_path_ok(self, features: List[torch.Tensor]) -> bool:
        return self._rows_path_reason(features) is None

    def _warn_stock_fallback(self, where: str, features: List[torch.Tensor]) -> None:
        """RES5_BACKEND "hip" was asked for and this call runs Res5 as torch.conv2d (MIOpen) instead: say so, once per call
        site and reason.  ("miopen" as the configured backend is a choice, not a fallback: silent.)"""
        if self.res5_backend != "hip":
            return
        reason = self._rows_path_reason(features)
        if reason is None and self._needs_graph(features) and self.res5_dtype not in ("fp32", "f16x2"):
            reason = f"RES5_DTYPE {self.res5_dtype!r} has no backward on the hand-written kernels (fp32 and f16x2 do)"
        if reason is None:
            return
        seen = self.__dict__.setdefault("_fallback_warned", set())
        if (where, reason) in seen:
            return
        seen.add((where, reason))
        import warnings
        warnings.warn(f"{type(self).__name__}.{where}: RES5_BACKEND is 'hip' but this call runs Res5 on torch.conv2d (the stock "
                      f"library path, ~8x slower): {reason}", RuntimeWarning, stacklevel=3)

    def _needs_graph(self, features: List[torch.Tensor]) -> bool:
        return torch.is_grad_enabled() and (features[0].requires_grad or any(p.requires_grad for p in self.res5.parameters()))

    def _fused_path_ok(self, features: List[torch.Tensor]) -> bool:
        """The inference form of the rows path (no autograd graph needed)."""
        return self._rows_path_ok(features) and not self._needs_graph(features)

    def _train_path_ok(self, features: List[torch.Tensor]) -> bool:
        """The differentiable form (locov_amd/res5_train.py): forward, data and weight gradients on the HIP kernels."""
        return self._rows_path_ok(features) and self.res5_dtype in ("fp32", "f16x2")

    def _res5_grid(self, feature: torch.Tensor, nhwc: Optional[torch.Tensor] = None) -> torch.Tensor:
        """roi_emb_heads.py:323: self.res5(features) on the whole res4 grid -> [N, C5, H/2, W/2]."""
        if not self._train_path_ok([feature]):
            self._warn_stock_fallback("_res5_grid", [feature])
            return self.res5(feature)
        from .. import res5_train
        if nhwc is None:
            nhwc = res5_train.to_nhwc(feature)
        return res5_train.res5_grid(self.res5, nhwc, split=self.res5_dtype == "f16x2", overflow_check=self.res5_overflow_check,
                                    on_overflow=self._warn_overflow)

    def _shared_roi_transform(self, features: List[torch.Tensor], boxes: List[Boxes], pooled: bool = False,
                              nhwc: Optional[torch.Tensor] = None, rois: Optional[torch.Tensor] = None):
        """roi_emb_heads.py:243-245: res5(pooler(features, boxes)) -> [R, C5, P/2, P/2]; with `pooled` the spatial
        mean of that tensor, [R, C5] (:262,:344,:356), which on the hand-written path in split arithmetic comes fused
        out of Res5's last 1x1 convolution.

        MI355X path: ROIAlign is evaluated on a channels-last copy of the map and only at the even
        bins the stride-2 1x1 convs of block 0 read; Res5 then runs as MFMA GEMMs over pixel rows.
        The result is returned as a logical NCHW tensor in channels-last memory."""
        if self._needs_graph(features) and self._train_path_ok(features):
            # training: the same pipeline as differentiable HIP ops (even-grid ROIAlign -> Res5 rows with data and
            # weight gradients -> mean); `nhwc` lets the caller share the channels-last copy with the whole-grid call
            from .. import res5_train
            assert len(boxes) == features[0].shape[0]
            if rois is None:                                 # (else: the pooler-format rows of `boxes`, already on the device)
                rois = convert_boxes_to_pooler_format(boxes)
            P = self.pooler.output_size[0]
            if nhwc is None:
                nhwc = res5_train.to_nhwc(features[0])
            R, o = rois.shape[0], P // 2
            y = res5_train.res5_rois(self.res5, nhwc, rois, P, self.pooler.scales[0], self.pooler.sampling_ratio, self.pooler.aligned,
                                     pooled=pooled, split=self.res5_dtype == "f16x2", overflow_check=self.res5_overflow_check,
                                     on_overflow=self._warn_overflow)
            return y if pooled else y.view(R, o, o, y.shape[1]).permute(0, 3, 1, 2)
        if not self._fused_path_ok(features):
            self._warn_stock_fallback("_shared_roi_transform", features)
            x = self.pooler(features, boxes)                 # :244
            x = self.res5(x)                                 # :245
            return self._pooled_mean(x) if pooled else x
        if self.res5_dtype == "f16x2" and self.res5_overflow_check:
            dev = features[0].device
            if ops.active_guard(dev) is not None:
                # the caller holds the guard (inference_detection / forward): the launches raise ITS word, it reads the word
                # once, behind everything it enqueued, and repeats the call on the f32 MFMA itself
                return self._fused_roi_transform(features, boxes, pooled, "f16x2")
            guard = self.res5.range_guard("fwd", dev)
            guard.reset()
            with ops.range_guard(guard):
                out = self._fused_roi_transform(features, boxes, pooled, "f16x2")
            if not guard.raised():                           # one 4-byte read per call
                return out
            self._warn_overflow()
            return self._fused_roi_transform(features, boxes, pooled, "fp32")
        return self._fused_roi_transform(features, boxes, pooled, self.res5_dtype)

    def _deferred_guard(self, feats: List[torch.Tensor]):
        """The range guard of a whole forward, read ONCE where the caller synchronises anyway -- or None when this call
        needs none (not the split arithmetic, check switched off, or not a hand-written path)."""
        if not (self.res5_dtype == "f16x2" and self.res5_overflow_check and self._rows_path_ok(feats)
                and hasattr(self.res5, "range_guard")):
            return None
        dev = feats[0].device
        if ops.active_guard(dev) is not None:
            return None                                      # an outer caller already holds one
        guard = self.res5.range_guard("fwd", dev)
        guard.reset()
        return guard

    def _deferred_guards(self, device):
        return self.res5.deferred_guards(device) if hasattr(self.res5, "deferred_guards") else []

    def _deferred_guards_tripped(self, kinds) -> None:
        if "bwd" in kinds:
            self._count("bwd_guard_trips")
            self._backward_guard_tripped()
        if "fwd_train" in kinds:
            self._count("guard_trips")
            self._count("deferred_trips")
            import warnings
            for kind, g in self.res5.deferred_guards(next(self.res5.parameters()).device):
                if kind == "fwd_train":
                    g.reset()
            if hasattr(self.res5, "forget_scales"):
                self.res5.forget_scales()
            self.res5_dtype = "fp32"
            warnings.warn("Res5 activations left the range of the f16x2 split arithmetic (|x| >= 4094) during the previous training "
                          "step: that step's Res5 outputs and gradients were ZEROED on the device (no inf / NaN reached the losses "
                          "or the optimizer; the step was skipped for the ROI-head path), and RES5_DTYPE is 'fp32' (the f32 MFMA) "
                          "from this step on.  MODEL.ROI_BOX_HEAD.RES5_TRAIN_GUARD 'sync' reads the guard inside every step and "
                          "repeats an out-of-range forward instead", RuntimeWarning, stacklevel=4)

    def _train_guard(self, feats: List[torch.Tensor]):
        """The DEFERRED range guard of a training forward (RES5_TRAIN_GUARD "deferred"; opt-in, the default is "sync":
        config.py, _deferred_guard): never read inside the
        step -- a host read behind the Res5 forward drains the GPU in front of the step's ~200 small launches (predictor,
        losses, grounding head) -- but acted on ON THE DEVICE: the forward's outputs and the backward's gradients are
        zero-filled when the word is set (ops.zero_if_raised), and the word travels with the next step's labelling read."""
        if not (self.res5_dtype == "f16x2" and self.res5_overflow_check and self.res5_train_guard == "deferred"
                and self._train_path_ok(feats) and self._needs_graph(feats) and hasattr(self.res5, "range_guard")):
            return None
        if ops.active_guard(feats[0].device) is not None:
            return None                                      # an outer caller already holds one
        g = self.res5.range_guard("fwd_train", feats[0].device)
        # THIS forward's copy of the word (filled by _close_train_guard at the end of the forward): what the backward of this
        # forward looks at -- a later forward's labelling read may clear the guard's own word before that backward runs
        # (gradient accumulation), it cannot reach this tensor
        g.step_word = torch.zeros(1, dtype=torch.int32, device=feats[0].device)
        return g

    @staticmethod
    def _close_train_guard(tguard, outputs, losses=None):
        """End of a forward under a DEFERRED guard: freeze the word for this forward, zero-fill the Res5 outputs when it is
        set, and return the factor (1 / 0, a device scalar) that turns this module's losses -- and with them every gradient
        they send into the predictor -- into a skipped step."""
        tguard.step_word.copy_(tguard.word)
        ops.zero_if_raised(list(outputs), tguard.step_word)    # (fresh contiguous tensors: zeroed in place, no host read)
        return (tguard.step_word == 0).to(torch.float32).reshape(())

    def _backward_guard_tripped(self) -> None:
        import warnings
        self.res5.backward_guard_tripped()
        warnings.warn("a remembered split-operand weight scale stopped covering its weight during the previous step's Res5 backward "
                      "(weights grew 8x within 64 steps): that backward's Res5 gradients were ZEROED on the device (the step was "
                      "skipped for the Res5 weights and the feature map, GradScaler-style; no inf / NaN was applied); the scales "
                      "are chosen afresh from here on", RuntimeWarning, stacklevel=3)

    def _warn_overflow(self):
        if hasattr(self.res5, "forget_scales"):
            self.res5.forget_scales()            # remembered operand scales may be what no longer fits
        if not self._overflow_warned:
            import warnings
            warnings.warn("Res5 activations left the range of the f16x2 split arithmetic (|x| >= 4094): this call was repeated "
                          "on the f32 MFMA (RES5_DTYPE 'fp32' avoids the retry; reported once per module)", RuntimeWarning, stacklevel=3)
            self._overflow_warned = True

    def _fused_roi_transform(self, features, boxes, pooled, res5_dtype):
        assert len(boxes) == features[0].shape[0]
        rois = convert_boxes_to_pooler_format(boxes)
        P = self.pooler.output_size[0]
        nhwc = ops.nchw_to_nhwc(features[0].detach())
        # position-major pixel rows [oh, ow, R, C]: a GEMM tile then holds one tile position of many
        # ROIs, which lets the 3x3 convolutions skip their zero-padding taps
        oh = ow = P // 2
        R = rois.shape[0]
        wino = self.res5_conv3x3 == "winograd"
        split = res5_dtype == "f16x2"
        if res5_dtype == "bf16":
            x0 = torch.empty((oh * ow * R, nhwc.shape[3]), dtype=torch.float32, device=nhwc.device)
            if P == 14 and self.res5[0].shortcut is not None:
                y = self.res5.forward_from_map(nhwc, rois, P, self.pooler.scales[0], self.pooler.sampling_ratio,
                                               self.pooler.aligned, bf16=True)
            else:
                ops.roi_align_nhwc(nhwc, rois, P, self.pooler.scales[0], self.pooler.sampling_ratio, self.pooler.aligned,
                                   bin_stride=2, pos_major=True, out=x0)
                y = self.res5.forward_rows(x0, oh, ow, pos_major=True, bf16=True)
            y = y.view(oh, ow, R, y.shape[1]).permute(2, 3, 0, 1)
            return self._pooled_mean(y) if pooled else y
        if P == 14 and self.res5.map_path_pays(R, nhwc.shape[0] * nhwc.shape[1] * nhwc.shape[2]):
            # many proposals per image: block 0's 1x1 convolutions run on the map, ROIAlign pools their outputs
            # (ROI-major rows with the Winograd form: its transforms and the mean-fused last convolution prefer a ROI's
            # 49 rows adjacent; the direct 3x3 form needs position-major rows for its tap skipping)
            y = self.res5.forward_from_map(nhwc, rois, P, self.pooler.scales[0], self.pooler.sampling_ratio,
                                           self.pooler.aligned, winograd=wino, split=split, pooled=pooled, roi_major=wino)
            if pooled:
                return y
            return y.view(R, oh, ow, y.shape[1]).permute(0, 3, 1, 2) if wino else y.view(oh, ow, R, y.shape[1]).permute(2, 3, 0, 1)
        # (ROIAlign writes straight into the operand block 0's K-concatenated conv3 + shortcut GEMM reads)
        x0 = self.res5.rows_input(oh * ow * R, nhwc.device)
        pm = not (wino and oh == 7)                      # ROI-major rows with the Winograd form, as on the map path
        ops.roi_align_nhwc(nhwc, rois, P, self.pooler.scales[0], self.pooler.sampling_ratio, self.pooler.aligned,
                           bin_stride=2, pos_major=pm, out=x0)
        y = self.res5.forward_rows(x0, oh, ow, pos_major=pm, winograd=wino, split=split, pooled=pooled)
        if pooled:
            return y
        # logical [R, C5, oh, ow]
        return y.view(oh, ow, R, y.shape[1]).permute(2, 3, 0, 1) if pm else y.view(R, oh, ow, y.shape[1]).permute(0, 3, 1, 2)

    def _pooled_mean(self, box_features: torch.Tensor) -> torch.Tensor:
        """box_features.mean(dim=[2,3]) (:262,:344,:356) on the HIP kernel."""
        if torch.is_grad_enabled() and box_features.requires_grad:
            return box_features.mean(dim=[2, 3])         # differentiable form for training
        if not box_features.is_contiguous():
            if box_features.permute(2, 3, 0, 1).is_contiguous():       # position-major [h,w,R,C]
                return ops.spatial_mean(box_features.permute(2, 3, 0, 1), channels_last=2)
            if box_features.permute(0, 2, 3, 1).is_contiguous():       # channels-last [R,h,w,C]
                return ops.spatial_mean(box_features.permute(0, 2, 3, 1), channels_last=1)
        return ops.spatial_mean(box_features)

    def forward(self, images, features, proposals, targets=None):
        """roi_emb_heads.py:247-282."""
        del images
        if not self.training:
            del targets
            return self._detect(features, proposals)
        assert targets
        # (:250 labels first.  Here the labelling's kernels are enqueued and nothing waits for them: the sample is formed on the
        # device on the assumption that every image fills its budget (_label_speculate), ROIAlign, Res5, the predictor and the
        # losses are enqueued behind it, and the step's ONE host wait -- the labelling's integers and the range-guard word --
        # comes at the end of this forward; a batch that did not fill its budget is repeated from the true counts)
        pending = self._label_begin(proposals, targets)
        del targets
        feats = [features[f] for f in self.in_features]
        del features

        def attempt(sampled):
            props = sampled if sampled is not None else self._label_finish(pending)
            proposal_boxes = [x.proposal_boxes for x in props]
            tguard = self._train_guard(feats)
            guard = None if tguard is not None else self._deferred_guard(feats)
            with ops.range_guard(tguard if tguard is not None else guard):
                box_features = self._shared_roi_transform(feats, proposal_boxes, pooled=True,
                                                          rois=pending["rois"] if pending.get("rois_of") is props else None)
            keep = None
            if tguard is not None:
                keep = self._close_train_guard(tguard, [box_features])
            elif guard is not None and guard.event is None:
                guard.snapshot()                             # 4 bytes to pinned memory + an event, behind the Res5 launches
            predictions = self.box_predictor(box_features)   # (:261-262: the mean is all the predictor sees)
            losses = self._predictor_losses(predictions, props)
            if keep is not None:
                losses = {k: v * keep for k, v in losses.items()}
            return props, losses, guard

        dtype_was = self.res5_dtype
        sampled = self._label_speculate(pending)
        self._count("forwards")
        self._count("speculated" if sampled is not None else "unspeculated")
        proposals, losses, guard = attempt(sampled)
        if sampled is not None:
            filled = self._label_validate(pending)
            if not (filled and self.res5_dtype == dtype_was):
                if not filled:
                    self._count("speculation_misses")
                del losses
                proposals, losses, guard = attempt(None)
        if guard is not None and guard.raised():             # RES5_TRAIN_GUARD "sync": waits for the event only -- the predictor's
            self._warn_overflow()                            # and the losses' launches stay queued behind it
            self._count("guard_trips")
            self._count("fp32_repeats")
            del losses
            proposals, losses, _ = self._with_res5_dtype("fp32", attempt, proposals)
        return [], losses

    def _predictor_losses(self, predictions, proposals):
        """box_predictor.losses on proposals that label_and_sample_proposals has validated (its one host read covers
        get_deltas' "Input boxes ... are not valid!" assert); a predictor without the keyword runs its own check."""
        import inspect
        fn = self.box_predictor.losses
        if "boxes_validated" in inspect.signature(fn).parameters:
            return fn(predictions, proposals, boxes_validated=True)
        return fn(predictions, proposals)

    def _predict(self, box_features, force_fp32: bool = False):
        """box_predictor(box_features); force_fp32 reaches predictors that know the keyword (this package's)."""
        if force_fp32:
            import inspect
            if "force_fp32" in inspect.signature(self.box_predictor.forward).parameters:
                return self.box_predictor(box_features, force_fp32=True)
        return self.box_predictor(box_features)

    def _with_res5_dtype(self, dtype, fn, *args, **kw):
        was, self.res5_dtype = self.res5_dtype, dtype
        try:
            return fn(*args, **kw)
        finally:
            self.res5_dtype = was

    def _detect(self, features, proposals):
        """The inference half of roi_emb_heads.py:247-282 / :351-360: ROIAlign + Res5 + mean -> box predictor -> post-processing.
        The split arithmetic's range guard costs NO host synchronisation here: the word is copied to pinned memory behind the
        predictor (4 bytes, asynchronous) and looked at after the post-processing, whose own host reads have by then waited
        for it; only a call that actually left the range is repeated (on the f32 MFMA) -- up to the post-processing the step
        holds no host sync, allocation-free and graph-capturable."""
        proposal_boxes = [x.proposal_boxes for x in proposals]
        feats = [features[f] for f in self.in_features]
        guard = self._deferred_guard(feats) if self._fused_path_ok(feats) else None
        with ops.range_guard(guard):
            box_features = self._shared_roi_transform(feats, proposal_boxes, pooled=True)     # :355-356
            predictions = self.box_predictor(box_features)
        if guard is not None:
            guard.snapshot()
        pred_instances, _ = self.box_predictor.inference(predictions, proposals)
        if guard is not None and guard.raised():
            self._warn_overflow()
            box_features = self._fused_roi_transform(feats, proposal_boxes, True, "fp32")
            predictions = self._predict(box_features, force_fp32=True)
            pred_instances, _ = self.box_predictor.inference(predictions, proposals)
        pred_instances = self.forward_with_given_boxes(features, pred_instances)
        return pred_instances, {}

    def forward_with_given_boxes(self, features, instances):
        """roi_emb_heads.py:284-306 (mask branch not part of this path)."""
        assert not self.training
        assert instances[0].has("pred_boxes") and instances[0].has("pred_classes")
        return instances


@ROI_HEADS_REGISTRY.register()
class EmbeddingProposalsRes5ROIHeads(EmbeddingRes5ROIHeads):
    """roi_emb_heads.py:310 -- the LSM variant: also returns the whole-grid Res5 features and
    the per-image region features for the grounding branch."""

    def forward(self, images, features, proposals, targets=None):
        del images
        if targets is None:                              # keyed on targets, not self.training (:316)
            return self.inference_detection(features, proposals)
        # :318 labels first, :323 then runs Res5 on the whole grid, :343 on the sampled proposals.  Here the labelling's kernels are
        # enqueued and NOTHING waits for them: the sample is formed on the device on the assumption that every image fills its
        # budget (_label_speculate), the grid call, ROIAlign, the proposals' call, the predictor and the losses are enqueued
        # behind it, and the ONE host wait of the step comes at the end of this forward -- the labelling's few integers (asserts,
        # counts, whether the budget was filled) and the range-guard word together.  A batch that did not fill its budget, or a
        # forward that left the split arithmetic's range, is repeated (from the true counts / on the f32 MFMA).
        pending = self._label_begin(proposals, targets)
        del targets
        feats = [features[f] for f in self.in_features]
        joint = self._train_path_ok(feats) and self._needs_graph(feats)
        nhwc = None
        if joint:
            from .. import res5_train
            nhwc = res5_train.to_nhwc(feats[0])          # one channels-last copy (and one gradient transpose) for both calls
        del features

        def attempt(sampled):
            """One forward of the step; sampled: the speculated sample, or None = wait for the labelling (behind the grid call)."""
            # ONE range guard for both Res5 calls (the whole grid and the sampled proposals).  "sync": one look at the word, at the
            # end; "deferred": acted on on the device, read with the next step's labelling
            tguard = self._train_guard(feats)
            guard = None if tguard is not None else self._deferred_guard(feats)
            with ops.range_guard(tguard if tguard is not None else guard):
                if joint:
                    grid, box, props = self._res5_both(feats[0], nhwc, pending, sampled)
                else:
                    grid = self._res5_grid(feats[0], nhwc)                           # :323
                    props = sampled if sampled is not None else self._label_finish(pending)      # :318
                    box = self._shared_roi_transform(feats, [x.proposal_boxes for x in props], pooled=True, nhwc=nhwc)   # :343-344
            keep = None
            if tguard is not None:
                keep = self._close_train_guard(tguard, [grid, box])
            elif guard is not None and guard.event is None:
                guard.snapshot()                             # 4 bytes to pinned memory + an event, behind both Res5 calls
            predictions = self.box_predictor(box)                                    # :345
            losses = dict(self._predictor_losses(predictions, props))                # :347
            if keep is not None:
                losses = {k: v * keep for k, v in losses.items()}
            return grid, box, props, losses, guard

        dtype_was = self.res5_dtype
        sampled = self._label_speculate(pending)
        self._count("forwards")
        self._count("speculated" if sampled is not None else "unspeculated")
        visual_grid_features, box_features, proposals, losses, guard = attempt(sampled)
        if sampled is not None:
            filled = self._label_validate(pending)
            if not (filled and self.res5_dtype == dtype_was):
                # the budget was not filled (the true counts are known now), or the read found the PREVIOUS step's deferred guard
                # set (RES5_DTYPE is "fp32" from here on): the forward enqueued above is dropped and repeated
                if not filled:
                    self._count("speculation_misses")
                del visual_grid_features, box_features, proposals, losses
                visual_grid_features, box_features, proposals, losses, guard = attempt(None)
        if guard is not None and guard.raised():
            # RES5_TRAIN_GUARD "sync": the wait is for the event in front of the stage's last convolution -- the launches behind
            # it (that convolution, the predictor, the losses) are still queued when the host returns to the caller, so the GPU
            # does not drain.  A step that left the split arithmetic's range is repeated on the f32 MFMA (the first graph is dropped)
            self._warn_overflow()
            self._count("guard_trips")
            self._count("fp32_repeats")
            del visual_grid_features, box_features, losses
            visual_grid_features, box_features, proposals, losses, _ = self._with_res5_dtype("fp32", attempt, proposals)
        box_features = list(box_features.split([len(x) for x in proposals], dim=0))      # :346
        return visual_grid_features, box_features, proposals, losses

    def _res5_both(self, feature: torch.Tensor, nhwc: torch.Tensor, pending, sampled=None):
        """roi_emb_heads.py:318-344 on the hand-written training path: self.res5(features) on the whole grid (:323) and
        res5(pooler(...)).mean() on the sampled proposals (:343-344) as the two segments of ONE res5_train.Res5Step; the
        backward of both is one joint pass over their 4 200 + 39 200 rows.  Without a speculated sample the grid call is enqueued
        first and the host then waits for the labelling's few integers (:318) with that call still queued.
        Returns (visual_grid_features [N, C5, H/2, W/2], pooled box_features [R, C5], sampled proposals)."""
        from .. import res5_train
        N, H, W, _ = nhwc.shape
        P = self.pooler.output_size[0]
        o = (P + 1) // 2
        held = ops.active_guard(nhwc.device)
        early = held.snapshot if held is not None and not getattr(held, "deferred", False) else None
        if sampled is not None and _JOINT_FORWARD:
            # the sample is known without a host wait: both calls are forwarded together (their 1x1 convolutions share launches)
            step = res5_train.Res5Step(self.res5, self.res5_dtype == "f16x2", nhwc.device,
                                       res5_train.grid_capacity(nhwc) + o * o * N * self.batch_size_per_image)
            rois = self._sampled_rois(pending, sampled)
            rows, x0 = res5_train.grid_and_roi_segments(step, nhwc, rois, P, self.pooler.scales[0], self.pooler.sampling_ratio,
                                                        self.pooler.aligned, on_range_final=early)
            grid, box_features = step.outputs([rows, x0], [False, True])
            return res5_train.to_nchw(grid, N, (H + 1) // 2, (W + 1) // 2), box_features, sampled
        while True:
            dtype = self.res5_dtype
            step = res5_train.Res5Step(self.res5, dtype == "f16x2", nhwc.device,
                                       res5_train.grid_capacity(nhwc) + o * o * N * self.batch_size_per_image)
            rows = res5_train.grid_segment(step, nhwc)
            proposals = sampled if sampled is not None else self._label_finish(pending)      # :318
            if self.res5_dtype == dtype:
                break
            # that read found the PREVIOUS step's deferred guard set: RES5_DTYPE is "fp32" from here on, and the grid call
            # enqueued above (still in split arithmetic, on data that may again be out of range) is redone
            del step, rows
        rois = convert_boxes_to_pooler_format([x.proposal_boxes for x in proposals])
        # (a guard the caller will WAIT for -- RES5_TRAIN_GUARD "sync" -- is snapshot in front of the stage's last convolution, the
        # first point at which nothing can raise it any more: ~0.3 ms of queued GPU work more for the host to come back to)
        x0 = res5_train.roi_segment(step, nhwc, rois, P, self.pooler.scales[0], self.pooler.sampling_ratio, self.pooler.aligned,
                                    on_range_final=early)
        grid, box_features = step.outputs([rows, x0], [False, True])
        return res5_train.to_nchw(grid, N, (H + 1) // 2, (W + 1) // 2), box_features, proposals

    def inference_detection(self, features, proposals):
        """roi_emb_heads.py:351-360."""
        return self._detect(features, proposals)


def build_roi_heads(cfg, input_shape: Dict[str, ShapeSpec]):
    """[D2-upstream] build_roi_heads: cfg.MODEL.ROI_HEADS.NAME -> registry."""
    return ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, input_shape)
