"""Box predictor of the LSM ROI head: bbox_pred FC, emb_pred FC, optional row normalisation and
the region x text similarity GEMM against the frozen noun bank -- on the gfx950 kernels.

Mirrors ovr/modeling/roi_heads/box_emb_head.py (class, method, attribute, config and
checkpoint-key names):
    EmbeddingFastRCNNOutputLayers.__init__            :69-149
    .from_config                                      :152-177
    .forward / .forward_cls_prediction                :179-212
    .set_class_embeddings                             :214-236
    build_box_predictor                               :239-249
and the parts of its Detectron2 base class the reference relies on ([D2-upstream]
FastRCNNOutputLayers: bbox_pred creation/init, losses, inference, predict_boxes,
predict_probs; Box2BoxTransform; fast_rcnn_inference).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Tuple, Union

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from .. import ops
from ..registry import configurable
from ..structures import Boxes, Instances, ShapeSpec, boxes_class_of, cat_rows

__all__ = ["Box2BoxTransform", "FastRCNNOutputLayers", "EmbeddingFastRCNNOutputLayers", "build_box_predictor",
           "fast_rcnn_inference", "batched_nms"]

_DEFAULT_SCALE_CLAMP = math.log(1000.0 / 16)
_FUSED_BOX_LOSS = os.environ.get("LOCOV_FUSED_LOSSES", "1") != "0"          # (developer A/B switch: tools/attic/ab_fused_losses.py)
_FUSED_POSTPROCESS = os.environ.get("LOCOV_FUSED_POSTPROCESS", "1") != "0"  # (developer A/B / tests: 0 = the torch-op post-processing chain)


class Box2BoxTransform:
    """[D2-upstream] R-CNN box parameterisation (dx, dy, dw, dh) with weights (10,10,5,5)."""

    def __init__(self, weights: Tuple[float, float, float, float], scale_clamp: float = _DEFAULT_SCALE_CLAMP):
        self.weights = tuple(float(w) for w in weights)
        self.scale_clamp = scale_clamp

    def get_deltas(self, src_boxes: torch.Tensor, target_boxes: torch.Tensor, check: bool = True) -> torch.Tensor:
        """check=False: the caller has already established that every source box has positive width and height (the ROI
        heads' labelling reads that bit with its one host read and raises this same AssertionError there)."""
        src_w = src_boxes[:, 2] - src_boxes[:, 0]
        src_h = src_boxes[:, 3] - src_boxes[:, 1]
        src_cx = src_boxes[:, 0] + 0.5 * src_w
        src_cy = src_boxes[:, 1] + 0.5 * src_h
        tgt_w = target_boxes[:, 2] - target_boxes[:, 0]
        tgt_h = target_boxes[:, 3] - target_boxes[:, 1]
        tgt_cx = target_boxes[:, 0] + 0.5 * tgt_w
        tgt_cy = target_boxes[:, 1] + 0.5 * tgt_h
        wx, wy, ww, wh = self.weights
        dx = wx * (tgt_cx - src_cx) / src_w
        dy = wy * (tgt_cy - src_cy) / src_h
        dw = ww * torch.log(tgt_w / src_w)
        dh = wh * torch.log(tgt_h / src_h)
        deltas = torch.stack((dx, dy, dw, dh), dim=1)
        if check:       # (a host read, as in Detectron2; device-side asserts compile to nothing on this ROCm build)
            assert bool(((src_w > 0) & (src_h > 0)).all()), "Input boxes to Box2BoxTransform are not valid!"
        return deltas

    def apply_deltas(self, deltas: torch.Tensor, boxes: torch.Tensor) -> torch.Tensor:
        deltas = deltas.float()
        boxes = boxes.to(deltas.dtype)
        widths = boxes[:, 2] - boxes[:, 0]
        heights = boxes[:, 3] - boxes[:, 1]
        ctr_x = boxes[:, 0] + 0.5 * widths
        ctr_y = boxes[:, 1] + 0.5 * heights
        wx, wy, ww, wh = self.weights
        dx = deltas[:, 0::4] / wx
        dy = deltas[:, 1::4] / wy
        dw = deltas[:, 2::4] / ww
        dh = deltas[:, 3::4] / wh
        dw = torch.clamp(dw, max=self.scale_clamp)
        dh = torch.clamp(dh, max=self.scale_clamp)
        pred_ctr_x = dx * widths[:, None] + ctr_x[:, None]
        pred_ctr_y = dy * heights[:, None] + ctr_y[:, None]
        pred_w = torch.exp(dw) * widths[:, None]
        pred_h = torch.exp(dh) * heights[:, None]
        x1 = pred_ctr_x - 0.5 * pred_w
        y1 = pred_ctr_y - 0.5 * pred_h
        x2 = pred_ctr_x + 0.5 * pred_w
        y2 = pred_ctr_y + 0.5 * pred_h
        return torch.stack((x1, y1, x2, y2), dim=-1).reshape(deltas.shape)


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """[D2-upstream] torchvision.ops.nms semantics (greedy, score-descending, IoU > thr suppressed).
    torchvision is not available on the ROCm box; the IoU matrix is built on the device and the
    greedy sweep (inherently sequential, a few hundred candidates) runs over its boolean mask."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.int64, device=boxes.device)
    if boxes.is_cuda:
        return ops.nms(boxes.float(), scores.float(), iou_threshold)     # HIP bit-matrix + on-device sweep
    # CPU tensors (host-logic unit tests only): same algorithm with a numpy sweep
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(b[:, None, :2], b[None, :, :2])
    rb = torch.min(b[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    iou = inter / (area[:, None] + area[None, :] - inter)
    over = (iou > iou_threshold).cpu().numpy()
    keep = []
    suppressed = np.zeros(n, dtype=bool)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        suppressed[i + 1:] |= over[i, i + 1:]
    return order[torch.as_tensor(keep, dtype=torch.int64, device=boxes.device)]


_PER_CLASS_NMS_ABOVE = 40000      # [D2-upstream] detectron2.layers.batched_nms: per-class loop from 40 000 candidates on


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """[D2-upstream] class-wise NMS.  Up to 40 000 candidates: ONE NMS over boxes shifted by per-class coordinate offsets
    (torchvision's batched_nms trick).  Above that -- 1203 classes x 1000 proposals with a low SCORE_THRESH_TEST reach
    1e5..1e6 candidates, whose K x K/64 suppression bit-matrix would take gigabytes -- the classes are independent, so each
    runs its own NMS and the kept indices are merged in descending score order, as Detectron2 does."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    if boxes.shape[0] < _PER_CLASS_NMS_ABOVE:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + 1)                    # (a python scalar: no host-to-device copy)
        return nms(boxes + offsets[:, None], scores, iou_threshold)
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for cid in torch.unique(idxs).tolist():
        sel = (idxs == cid).nonzero().view(-1)
        keep_mask[sel[nms(boxes[sel], scores[sel], iou_threshold)]] = True
    keep = keep_mask.nonzero().view(-1)
    return keep[scores[keep].argsort(descending=True, stable=True)]


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh: float, nms_thresh: float,
                                     topk_per_image: int, instances_cls=Instances, boxes_cls=Boxes, all_finite: bool = False):
    """all_finite: the caller has already established that every box / score of the batch is finite (one host read for the
    whole batch instead of one per image)."""
    valid_mask = None if all_finite else torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    if valid_mask is not None and not valid_mask.all():
        import warnings
        warnings.warn(f"fast_rcnn_inference: dropping {int((~valid_mask).sum())} proposals with non-finite boxes / scores "
                      "(as Detectron2 does) -- the ROI head produced inf / NaN for them", RuntimeWarning, stacklevel=2)
        boxes, scores = boxes[valid_mask], scores[valid_mask]
    scores = scores[:, :-1]
    num_bbox_reg_classes = boxes.shape[1] // 4
    boxes = Boxes(boxes.reshape(-1, 4))
    boxes.clip(image_shape)
    boxes = boxes.tensor.view(-1, num_bbox_reg_classes, 4)
    filter_mask = scores > score_thresh
    filter_inds = filter_mask.nonzero()
    if num_bbox_reg_classes == 1:
        boxes = boxes[filter_inds[:, 0], 0]
    else:
        boxes = boxes[filter_mask]
    scores = scores[filter_mask]
    keep = batched_nms(boxes, scores, filter_inds[:, 1], nms_thresh)
    if topk_per_image >= 0:
        keep = keep[:topk_per_image]
    boxes, scores, filter_inds = boxes[keep], scores[keep], filter_inds[keep]
    result = instances_cls(image_shape)
    result.pred_boxes = boxes_cls(boxes)
    result.scores = scores
    result.pred_classes = filter_inds[:, 1]
    return result, filter_inds[:, 0]


def _fast_rcnn_inference_batched(all_boxes: torch.Tensor, all_scores: torch.Tensor, sizes: List[int], image_shapes,
                                 score_thresh: float, nms_thresh: float, topk_per_image: int, instances_cls, boxes_cls):
    """fast_rcnn_inference for a whole batch (concatenated device tensors, `sizes` rows per image) with class-agnostic boxes: the same arithmetic per image as
    fast_rcnn_inference_single_image (clip, score filter, class-wise NMS on boxes shifted by class * (max coordinate of the
    image's candidates + 1), top-k in descending score order), but every elementwise / sort / scan step runs ONCE over all
    images' candidates.  The per-image form launches ~40 small kernels and waits for the host five times per image (8 images:
    ~320 launches, 4 ms at 80 classes); this one needs ~50 launches and three host reads per batch.
    Returns None when a case it does not cover shows up (>= 40 000 candidates in an image: per-class NMS)."""
    dev = all_boxes.device                                                # all_boxes [R, 4], all_scores [R, K + 1] (all images)
    B = len(sizes)
    all_scores = all_scores[:, :-1]
    # per-row image index and clip limits, built by fill kernels from host-side sizes (no host-to-device copy)
    img_of_row = torch.empty((all_boxes.shape[0],), dtype=torch.int64, device=dev)
    lim = torch.empty((all_boxes.shape[0], 2), dtype=all_boxes.dtype, device=dev)
    r0 = 0
    for i, (n, (h, w)) in enumerate(zip(sizes, image_shapes)):
        img_of_row[r0:r0 + n] = i
        lim[r0:r0 + n, 0] = float(w)
        lim[r0:r0 + n, 1] = float(h)
        r0 += n
    zero = all_boxes.new_zeros(())
    clipped = torch.stack((torch.minimum(torch.maximum(all_boxes[:, 0], zero), lim[:, 0]),
                           torch.minimum(torch.maximum(all_boxes[:, 1], zero), lim[:, 1]),
                           torch.minimum(torch.maximum(all_boxes[:, 2], zero), lim[:, 0]),
                           torch.minimum(torch.maximum(all_boxes[:, 3], zero), lim[:, 1])), dim=-1)      # Boxes.clip
    cand = (all_scores > score_thresh).nonzero()                          # host read 1: [C, 2] (row, class), row-major
    rows, cls = cand[:, 0], cand[:, 1]
    img = img_of_row[rows]
    per_img = torch.bincount(img, minlength=B).tolist()                   # host read 2: candidates per image
    if max(per_img, default=0) >= _PER_CLASS_NMS_ABOVE:
        return None
    c_scores = all_scores[rows, cls]
    c_boxes = clipped[rows]
    # order: image-major, inside an image descending score, ties in candidate (row, class) order -- two stable sorts
    o1 = torch.argsort(c_scores, descending=True, stable=True)
    order = o1[torch.argsort(img[o1], stable=True)]
    s_boxes, s_scores, s_cls, s_img, s_rows = c_boxes[order], c_scores[order], cls[order], img[order], rows[order]
    # class offsets of batched_nms, per image: class * (max coordinate among THAT image's candidates + 1)
    max_coord = torch.full((B,), float("-inf"), dtype=s_boxes.dtype, device=dev).scatter_reduce(
        0, s_img, s_boxes.amax(dim=1), reduce="amax", include_self=True)
    shifted = (s_boxes + (s_cls.to(s_boxes) * (max_coord[s_img] + 1))[:, None]).contiguous()
    keep = torch.empty((shifted.shape[0],), dtype=torch.uint8, device=dev)
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    lib = ops._lib.load()
    c0 = 0
    with torch.cuda.device(dev):
        for i, n in enumerate(per_img):                                   # one greedy NMS per image on its (sorted) segment
            if n:
                ws = ops._workspace("nms", shifted, int(lib.locov_nms_workspace_bytes(n)))
                ops.check(lib.locov_nms_sorted(shifted[c0:c0 + n].data_ptr(), n, float(nms_thresh), ws.data_ptr(),
                                               keep[c0:c0 + n].data_ptr(), num[i:i + 1].data_ptr(), ops._stream(shifted)),
                          "locov_nms_sorted")
            c0 += n
    keep = keep.bool()
    if topk_per_image >= 0:                                               # rank of every kept candidate inside its image
        csum = torch.cumsum(keep.to(torch.int64), dim=0)
        seg_start = torch.zeros((B,), dtype=torch.int64, device=dev)
        c0 = 0
        for i, n in enumerate(per_img):                                   # (fill kernels from host-side counts)
            seg_start[i] = c0
            c0 += n
        before = torch.where(seg_start > 0, csum[(seg_start - 1).clamp(min=0)], torch.zeros_like(seg_start)) if keep.numel() else seg_start
        keep = keep & ((csum - before[s_img]) <= topk_per_image)
    final = keep.nonzero().view(-1)                                       # host read 3 (image-major, score-descending)
    counts = torch.bincount(s_img[final], minlength=B).tolist()
    f_boxes, f_scores, f_cls, f_rows = s_boxes[final], s_scores[final], s_cls[final], s_rows[final]
    results, kept_rows = [], []
    c0 = r0 = 0
    for i, n in enumerate(counts):
        res = instances_cls(image_shapes[i])
        res.pred_boxes = boxes_cls(f_boxes[c0:c0 + n])
        res.scores = f_scores[c0:c0 + n]
        res.pred_classes = f_cls[c0:c0 + n]
        results.append(res)
        kept_rows.append(f_rows[c0:c0 + n] - r0)
        c0 += n
        r0 += sizes[i]
    return results, kept_rows


def fast_rcnn_inference(boxes: List[torch.Tensor], scores: List[torch.Tensor], image_shapes, score_thresh: float,
                        nms_thresh: float, topk_per_image: int, instances_cls=Instances, boxes_cls=Boxes):
    """instances_cls / boxes_cls: the classes the results are built with (the caller's own -- Detectron2's under
    train_ovnet.py -- see structures.boxes_class_of)."""
    # Detectron2's finite-ness filter, evaluated once for the batch: per image it is a host sync each
    finite = True
    if len(boxes):
        cat_b, cat_s = torch.cat(list(boxes), dim=0), torch.cat(list(scores), dim=0)
        finite = bool(torch.isfinite(cat_b).all() & torch.isfinite(cat_s).all())
        if finite and cat_b.is_cuda and cat_b.shape[1] == 4:
            out = _fast_rcnn_inference_batched(cat_b, cat_s, [b.shape[0] for b in boxes], image_shapes, score_thresh, nms_thresh,
                                               topk_per_image, instances_cls, boxes_cls)
            if out is not None:
                return out
    result_per_image = [
        fast_rcnn_inference_single_image(b, s, shape, score_thresh, nms_thresh, topk_per_image, instances_cls, boxes_cls, all_finite=finite)
        for s, b, shape in zip(scores, boxes, image_shapes)
    ]
    return [x[0] for x in result_per_image], [x[1] for x in result_per_image]


def smooth_l1_loss(input: torch.Tensor, target: torch.Tensor, beta: float, reduction: str = "none"):
    """[fvcore] smooth_l1_loss; beta < 1e-5 -> plain L1."""
    if beta < 1e-5:
        loss = torch.abs(input - target)
    else:
        n = torch.abs(input - target)
        loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    if reduction == "mean":
        return loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
    if reduction == "sum":
        return loss.sum()
    return loss


def hip_linear(x: torch.Tensor, layer: nn.Linear) -> torch.Tensor:
    """nn.Linear forward (and backward) on the hand-written f32 MFMA kernel; the weights are read at
    call time (emb_pred.weight/bias are re-assigned by the meta-arch, distill_prop_mmss_gcnn.py:121-125)."""
    return ops.linear_autograd(x, layer.weight, layer.bias)


class FastRCNNOutputLayers(nn.Module):
    """[D2-upstream] the parts of FastRCNNOutputLayers the reference inherits: the bbox_pred
    layer, losses(), inference(), predict_boxes(), predict_probs()."""

    @configurable
    def __init__(self, input_shape, *, box2box_transform, num_classes: int, test_score_thresh: float = 0.0,
                 test_nms_thresh: float = 0.5, test_topk_per_image: int = 100, cls_agnostic_bbox_reg: bool = False,
                 smooth_l1_beta: float = 0.0, box_reg_loss_type: str = "smooth_l1",
                 loss_weight: Union[float, Dict[str, float]] = 1.0):
        super().__init__()
        if isinstance(input_shape, int):
            input_shape = ShapeSpec(channels=input_shape)
        self.num_classes = num_classes
        input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        self.cls_score = nn.Linear(input_size, num_classes + 1)
        num_bbox_reg_classes = 1 if cls_agnostic_bbox_reg else num_classes
        box_dim = len(box2box_transform.weights)
        self.bbox_pred = nn.Linear(input_size, num_bbox_reg_classes * box_dim)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for l in [self.cls_score, self.bbox_pred]:
            nn.init.constant_(l.bias, 0)
        self.box2box_transform = box2box_transform
        self.smooth_l1_beta = smooth_l1_beta
        self.test_score_thresh = test_score_thresh
        self.test_nms_thresh = test_nms_thresh
        self.test_topk_per_image = test_topk_per_image
        self.box_reg_loss_type = box_reg_loss_type
        if isinstance(loss_weight, float):
            loss_weight = {"loss_cls": loss_weight, "loss_box_reg": loss_weight}
        self.loss_weight = dict(loss_weight)

    @classmethod
    def from_config(cls, cfg, input_shape):
        return {
            "input_shape": input_shape,
            "box2box_transform": Box2BoxTransform(weights=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS),
            "num_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES,
            "cls_agnostic_bbox_reg": cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG,
            "smooth_l1_beta": cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA,
            "test_score_thresh": cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
            "test_nms_thresh": cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST,
            "test_topk_per_image": cfg.TEST.DETECTIONS_PER_IMAGE,
            "box_reg_loss_type": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE,
            "loss_weight": {"loss_box_reg": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT},
        }

    def forward(self, x):
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        return hip_linear(x, self.cls_score), hip_linear(x, self.bbox_pred)

    def losses(self, predictions, proposals, boxes_validated: bool = False):
        """boxes_validated: the proposals come from SampleAllROIHeads.label_and_sample_proposals, which has already
        checked (and raised for) degenerate foreground boxes -- get_deltas' host-side assert is then skipped."""
        scores, proposal_deltas = predictions
        # (cat_rows: the sampled Instances of a step are per-image views of batch-wide tensors -- their concatenation is a view)
        gt_classes = (cat_rows([p.gt_classes for p in proposals]) if len(proposals)
                      else torch.empty(0, dtype=torch.int64, device=scores.device))
        if len(proposals):
            proposal_boxes = cat_rows([p.proposal_boxes.tensor for p in proposals])
            assert not proposal_boxes.requires_grad, "Proposals should not require gradients!"
            gt_boxes = cat_rows([(p.gt_boxes if p.has("gt_boxes") else p.proposal_boxes).tensor for p in proposals])
        else:
            proposal_boxes = gt_boxes = torch.empty((0, 4), device=proposal_deltas.device)
        if gt_classes.numel() == 0:
            loss_cls = scores.sum() * 0.0
        else:
            loss_cls = F.cross_entropy(scores, gt_classes, reduction="mean")
        losses = {"loss_cls": loss_cls,
                  "loss_box_reg": self.box_reg_loss(proposal_boxes, gt_boxes, proposal_deltas, gt_classes,
                                                    boxes_validated=boxes_validated)}
        # (a weight of exactly 1 changes no bit: no launch for it, forward or backward)
        return {k: v if self.loss_weight.get(k, 1.0) == 1.0 else v * self.loss_weight[k] for k, v in losses.items()}

    def box_reg_loss(self, proposal_boxes, gt_boxes, pred_deltas, gt_classes, boxes_validated: bool = False):
        """[D2-upstream] smooth-L1 over the foreground rows, normalised by ALL rows.  The foreground rows are selected by a
        mask instead of `nonzero` indices (a data-dependent size = a host sync): background rows get a unit box as source and
        target (zero deltas, finite everywhere) and weight 0 in the sum -- same value, same gradients."""
        box_dim = proposal_boxes.shape[1]
        if self.box_reg_loss_type != "smooth_l1":
            raise ValueError(f"Invalid bbox reg loss type '{self.box_reg_loss_type}'")
        if (_FUSED_BOX_LOSS and boxes_validated and pred_deltas.is_cuda and box_dim == 4 and pred_deltas.dtype == torch.float32
                and type(self.box2box_transform) is Box2BoxTransform and gt_classes.numel() > 0):
            # one launch (ops.box_reg_loss = locov_box_reg_loss): the chain below, fused -- same foreground rule, same deltas, same
            # normalisation; the gradient of the predictions comes out of the same launch
            return ops.box_reg_loss(pred_deltas, proposal_boxes, gt_boxes, gt_classes, self.num_classes,
                                    self.box2box_transform.weights, self.smooth_l1_beta)
        fg = (gt_classes >= 0) & (gt_classes < self.num_classes)
        if pred_deltas.shape[1] == box_dim:
            pred = pred_deltas
        else:
            rows = torch.arange(gt_classes.shape[0], device=gt_classes.device)
            pred = pred_deltas.view(-1, self.num_classes, box_dim)[rows, gt_classes.clamp(0, self.num_classes - 1)]
        unit = torch.cat([proposal_boxes.new_zeros(box_dim // 2), proposal_boxes.new_ones(box_dim - box_dim // 2)])   # [0, 0, 1, 1], built on the device
        src_boxes = torch.where(fg[:, None], proposal_boxes, unit)        # (background rows may be degenerate boxes: they
        target_boxes = torch.where(fg[:, None], gt_boxes, unit)           # never reach get_deltas' validity check upstream)
        gt_pred_deltas = self.box2box_transform.get_deltas(src_boxes, target_boxes, check=not boxes_validated)
        # background / ignored rows are REPLACED, not multiplied by zero: a non-finite prediction in such a row (which the
        # indexed upstream form never touches) must not reach the value (0 * inf = NaN) or the gradient
        pred = torch.where(fg[:, None], pred, gt_pred_deltas)
        per_elem = smooth_l1_loss(pred, gt_pred_deltas, self.smooth_l1_beta, reduction="none")
        loss_box_reg = torch.where(fg[:, None], per_elem, per_elem.new_zeros(())).sum()
        return loss_box_reg / max(gt_classes.numel(), 1.0)

    def inference(self, predictions, proposals):
        fused = self._inference_fused(predictions, proposals)
        if fused is not None:
            return fused
        boxes = self.predict_boxes(predictions, proposals)
        scores = self.predict_probs(predictions, proposals)
        image_shapes = [x.image_size for x in proposals]
        instances_cls, boxes_cls = boxes_class_of(proposals)
        return fast_rcnn_inference(boxes, scores, image_shapes, self.test_score_thresh, self.test_nms_thresh,
                                   self.test_topk_per_image, instances_cls, boxes_cls)

    def _inference_fused(self, predictions, proposals):
        """predict_boxes + predict_probs + fast_rcnn_inference as ONE device pipeline (ops.detect_postprocess: csrc/detect.hip)
        for the case both reference configurations evaluate: class-agnostic box regression on device fp32 tensors, a top-k.
        Returns None for anything else -- and when the kernels flag non-finite values or too many candidates -- so that the
        caller runs the torch chain; the detections are bit-identical to that chain's (tests/test_gpu_postprocess.py)."""
        if not _FUSED_POSTPROCESS or not len(proposals):
            return None
        scores, deltas = predictions
        sizes = [len(p) for p in proposals]
        K = scores.shape[1] - 1
        if not (scores.is_cuda and scores.dtype == torch.float32 and deltas.dtype == torch.float32 and deltas.dim() == 2 and deltas.shape[1] == 4
                and scores.shape[0] == sum(sizes) > 0 and len(sizes) <= ops.DETECT_MAX_IMAGES and 1 <= K <= ops.DETECT_MAX_CLASSES
                and 1 <= self.test_topk_per_image <= ops.DETECT_MAX_TOPK and max(sizes) <= ops.DETECT_MAX_ROWS_PER_IMAGE):
            return None
        pieces = [p.proposal_boxes.tensor for p in proposals]
        if any(t.dtype != torch.float32 or not t.is_cuda for t in pieces):
            return None
        image_shapes = [x.image_size for x in proposals]
        probs = F.softmax(scores, dim=-1)                                  # (torch's own softmax: predict_probs' values)
        out = ops.detect_postprocess(probs, deltas, cat_rows(pieces), sizes, image_shapes, self.box2box_transform.weights,
                                     self.box2box_transform.scale_clamp, self.test_score_thresh, self.test_nms_thresh,
                                     self.test_topk_per_image)
        if out is None:
            return None
        boxes, det_scores, classes, rows, counts = out
        instances_cls, boxes_cls = boxes_class_of(proposals)
        results, kept = [], []
        for i, n in enumerate(counts):
            res = instances_cls(image_shapes[i])
            res.pred_boxes = boxes_cls(boxes[i, :n])
            res.scores = det_scores[i, :n]
            res.pred_classes = classes[i, :n]
            results.append(res)
            kept.append(rows[i, :n])
        return results, kept

    def predict_boxes(self, predictions, proposals):
        if not len(proposals):
            return []
        _, proposal_deltas = predictions
        num_prop_per_image = [len(p) for p in proposals]
        proposal_boxes = torch.cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        predict_boxes = self.box2box_transform.apply_deltas(proposal_deltas, proposal_boxes)
        return predict_boxes.split(num_prop_per_image)

    def predict_probs(self, predictions, proposals):
        scores, _ = predictions
        num_inst_per_image = [len(p) for p in proposals]
        probs = F.softmax(scores, dim=-1)
        return probs.split(num_inst_per_image, dim=0)


class EmbeddingFastRCNNOutputLayers(FastRCNNOutputLayers):
    """ovr/modeling/roi_heads/box_emb_head.py:60 -- same constructor arguments, attributes
    (emb_pred, bbox_pred, cls_score, embedding_based, emb_dim, num_classes, normalize_emb,
    standardize_emb, detach_cls_predictor, loss_weight) and state-dict keys."""

    @configurable
    def __init__(self, input_shape, *, box2box_transform, num_classes: int, test_score_thresh: float = 0.0,
                 test_nms_thresh: float = 0.5, test_topk_per_image: int = 100, cls_agnostic_bbox_reg: bool = False,
                 smooth_l1_beta: float = 0.0, box_reg_loss_type: str = "smooth_l1",
                 loss_weight: Union[float, Dict[str, float]] = 1.0, emb_dim: int = 768,
                 embedding_based: bool = True, freeze_emb_pred: bool = True, normalize_emb: bool = False,
                 standardize_emb: bool = False, detach_cls_predictor: bool = False, sim_gemm_dtype: str = "fp32",
                 fc_dtype: str = "fp32"):
        FastRCNNOutputLayers.__init__(
            self, input_shape, box2box_transform=box2box_transform, num_classes=num_classes,
            test_score_thresh=test_score_thresh, test_nms_thresh=test_nms_thresh,
            test_topk_per_image=test_topk_per_image, cls_agnostic_bbox_reg=cls_agnostic_bbox_reg,
            smooth_l1_beta=smooth_l1_beta, box_reg_loss_type=box_reg_loss_type, loss_weight=loss_weight)
        if isinstance(input_shape, int):
            input_shape = ShapeSpec(channels=input_shape)
        num_inputs = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        self.embedding_based = embedding_based
        assert sim_gemm_dtype in ("fp32", "bf16") and fc_dtype in ("fp32", "f16x2")
        self.sim_gemm_dtype = sim_gemm_dtype
        # extension: arithmetic of emb_pred and (when SIM_GEMM_DTYPE is "fp32") cls_score in inference -- "f16x2": fp32 in / fp32
        # out with the products formed from split (hi, lo) f16 operand pairs on the f16 matrix pipe, as Res5's GEMMs under
        # RES5_DTYPE "f16x2" (csrc/gemm_split.hip: error vs fp64 no larger than the f32 MFMA's); follows RES5_DTYPE
        self.fc_dtype = fc_dtype
        self._split_cache = {}
        self._bank_bf16 = None          # packed copy of cls_score.weight for the bf16 MFMA path
        if self.embedding_based:
            self.normalize_emb = normalize_emb
            self.standardize_emb = standardize_emb
            self.emb_dim = emb_dim
            self.emb_pred = nn.Linear(num_inputs, self.emb_dim)
            nn.init.normal_(self.emb_pred.weight, mean=0, std=0.01)       # :135
            nn.init.constant_(self.emb_pred.bias, 0)                      # :136
            assert cls_agnostic_bbox_reg
            # forward() can't be used until set_class_embeddings() has run (:138-140)
            self.num_classes = None
            self.cls_score = None
            if freeze_emb_pred:
                self.emb_pred.weight.requires_grad = False
                self.emb_pred.bias.requires_grad = False
        self.detach_cls_predictor = detach_cls_predictor
        if self.detach_cls_predictor:
            self.loss_weight.update({"loss_cls": 0.0})                    # :147-149

    @classmethod
    def from_config(cls, cfg, input_shape):
        ret = FastRCNNOutputLayers.from_config.__func__(cls, cfg, input_shape)
        box_head = cfg.MODEL.ROI_BOX_HEAD
        ret.update({
            "emb_dim": box_head.EMB_DIM,
            "embedding_based": box_head.EMBEDDING_BASED,
            "freeze_emb_pred": box_head.FREEZE_EMB_PRED,
            "normalize_emb": box_head.NORMALIZE_EMB_PRED,
            "standardize_emb": box_head.STANDARDIZE_EMB_PRED,
            "detach_cls_predictor": cfg.MODEL.ROI_HEADS.DETACH_CLASS_PREDICTOR,
            "sim_gemm_dtype": box_head.get("SIM_GEMM_DTYPE", "fp32") if hasattr(box_head, "get") else "fp32",
        })
        res5_dtype = box_head.get("RES5_DTYPE", "f16x2") if hasattr(box_head, "get") else "f16x2"
        backend = box_head.get("RES5_BACKEND", "hip") if hasattr(box_head, "get") else "hip"
        ret["fc_dtype"] = "f16x2" if (res5_dtype == "f16x2" and backend == "hip") else "fp32"
        return ret

    # ------------------------------------------------------------------ forward (:179-212)
    def _norm_mode(self) -> int:
        if self.normalize_emb and self.standardize_emb:
            return -1       # both: apply in sequence (the reference applies normalise then standardise)
        if self.normalize_emb:
            return ops.NORM_L2
        if self.standardize_emb:
            return ops.NORM_STANDARDIZE
        return ops.NORM_NONE

    def _split_weight(self, tag: str, w: torch.Tensor):
        """Split-operand packing of a weight read at call time (emb_pred.weight is re-assigned by the meta-arch, the bank by
        set_class_embeddings): re-packed whenever the tensor is replaced or modified in place."""
        # The entry KEEPS the tensor and compares by identity: a (data_ptr, _version) key alone can match a different tensor --
        # every bank swap builds a fresh nn.Linear (version 0 again) and the caching allocator may hand the freed bank's address
        # to the next one of the same shape (Res5Stage._split does the same).
        # (... and the storage address stays in the key: `w.data = other` swaps the storage under the SAME object and version)
        hit = self._split_cache.get(tag)
        if hit is None or hit[0] is not w or hit[1] != (w._version, w.data_ptr()):
            hit = self._split_cache[tag] = (w, (w._version, w.data_ptr()), ops.split_pack(w.detach().contiguous()))
        return hit[2]

    def _fc_split_ok(self, x: torch.Tensor) -> bool:
        return (self.fc_dtype == "f16x2" and x.is_cuda and x.dim() == 2 and x.shape[1] % 32 == 0 and self.emb_dim % 32 == 0
                and self.cls_score is not None and self.cls_score.weight.shape[0] % 4 == 0)

    @torch.no_grad()
    def region_embedding(self, x: torch.Tensor, force_fp32: bool = False) -> torch.Tensor:
        """emb_pred(x) (:206) in the arithmetic forward() uses in inference, before any normalisation: [R, C5] -> [R, D]."""
        x = x.detach()
        if self._fc_split_ok(x) and not force_fp32:
            return ops.linear_split(x, self._split_weight("emb", self.emb_pred.weight), self.emb_pred.bias.detach(), x_scale=16.0)
        return ops.linear(x, self.emb_pred.weight.detach(), self.emb_pred.bias.detach())

    def forward(self, x, force_fp32: bool = False):
        """x: per-region features [R, ...] (flattened like the reference, :189-190).
        Returns (scores [R,K+1], proposal_deltas [R,4]).  force_fp32: keep emb_pred / cls_score on the f32 MFMA for this call
        (the ROI heads' repeat of a call whose activations left the split arithmetic's range)."""
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        if not self.embedding_based:
            return hip_linear(x, self.cls_score), hip_linear(x, self.bbox_pred)
        if self.cls_score is None:
            raise RuntimeError("set_class_embeddings() must be called before forward() (box_emb_head.py:138-140)")
        needs_grad = torch.is_grad_enabled() and (
            x.requires_grad or self.bbox_pred.weight.requires_grad
            or (not self.detach_cls_predictor and self.emb_pred.weight.requires_grad))
        mode = self._norm_mode()
        if not needs_grad and mode >= 0 and not force_fp32 and self._fc_split_ok(x):
            # inference in split arithmetic (3 f16 MFMAs per fp32 product block instead of 16 f32 ones; fp32-level accuracy):
            # operand scale 16 for the pooled features (their range is Res5's, covered by its range guard), 1 for the
            # embedding (|emb| < 65504, full 22 bits down to 0.125, absolute floor 3e-8 below: its entries are small)
            x = x.detach()
            deltas = ops.linear(x, self.bbox_pred.weight.detach(), self.bbox_pred.bias.detach())
            emb = self.region_embedding(x)
            if mode != ops.NORM_NONE:
                emb = ops.rownorm(emb, mode)
            if self.sim_gemm_dtype == "bf16":
                scores = ops.sim_gemm_bf16(ops.to_bf16(emb), self._packed_bank())
            else:
                scores = ops.linear_split(emb, self._split_weight("bank", self.cls_score.weight), None, x_scale=1.0)
            if not self._cls_bias_is_zero():
                scores = scores + self.cls_score.bias.detach()
            return scores, deltas
        if not needs_grad and mode >= 0:
            # inference: one C call -> bbox_pred, emb_pred, (norm,) similarity GEMM launches
            x = x.detach()
            _, deltas, _, scores = ops.box_head(
                x, self.emb_pred.weight.detach(), self.emb_pred.bias.detach(), self.bbox_pred.weight.detach(),
                self.bbox_pred.bias.detach(), self.cls_score.weight, self._packed_bank(), mode,
                ops.BF16 if self.sim_gemm_dtype == "bf16" else ops.F32)
            if not self._cls_bias_is_zero():
                # set_class_embeddings zeroes the bias (:234); a loaded / edited one is honoured like the reference's
                # nn.Linear does (the similarity kernel itself carries no bias term)
                scores = scores + self.cls_score.bias.detach()
            return scores, deltas
        if self.detach_cls_predictor:                                     # :197-201
            proposal_deltas = hip_linear(x, self.bbox_pred)               # :196
            with torch.no_grad():
                scores = self.forward_cls_prediction(x.detach())
            return scores, proposal_deltas
        # both FCs read x: one autograd node, one backward call (locov_pool_fc_bwd) that accumulates grad_x across them
        emb, proposal_deltas = ops.pool_fc_autograd(x, self.emb_pred.weight, self.emb_pred.bias, self.bbox_pred.weight,
                                                    self.bbox_pred.bias)      # :196, :206
        return self.forward_cls_prediction(x, emb=emb), proposal_deltas

    def forward_cls_prediction(self, x, emb=None):                        # :204-212 (emb: emb_pred(x) when already formed)
        if self.embedding_based:
            x = hip_linear(x, self.emb_pred) if emb is None else emb
            if self.normalize_emb:
                x = _rownorm(x, ops.NORM_L2)
            if self.standardize_emb:
                x = _rownorm(x, ops.NORM_STANDARDIZE)
        if self.sim_gemm_dtype == "bf16" and not (torch.is_grad_enabled() and x.requires_grad):
            scores = ops.sim_gemm_bf16(ops.to_bf16(x), self._packed_bank())
            return scores if self._cls_bias_is_zero() else scores + self.cls_score.bias.detach()
        if self.embedding_based:
            return ops.sim_gemm_autograd(x, self.cls_score.weight, self.cls_score.bias)      # :211
        return hip_linear(x, self.cls_score)

    def _bank_key(self):
        # (the tensor object itself is part of the key -- held, so its id cannot be recycled -- next to its version: see _split_weight)
        w = self.cls_score.weight
        return (w, (w._version, w.data_ptr()))

    def _packed_bank(self):
        """bf16 copy of the bank for the bf16 MFMA similarity GEMM, re-packed whenever cls_score.weight is re-assigned,
        loaded or modified in place (keyed on the tensor's storage and version, like Res5Stage._packed)."""
        if self.sim_gemm_dtype != "bf16":
            return None
        key = self._bank_key()
        have = getattr(self, "_bank_bf16_key", None)
        if self._bank_bf16 is None or have is None or have[0] is not key[0] or have[1] != key[1]:
            self._bank_bf16 = ops.to_bf16(self.cls_score.weight.detach())
            self._bank_bf16_key = key
        return self._bank_bf16

    def install_packed_bank(self, bank_bf16: torch.Tensor) -> None:
        """A pre-packed bf16 copy of the CURRENT cls_score.weight (TextBankCache.install: no per-swap conversion)."""
        assert bank_bf16.dtype == torch.bfloat16 and tuple(bank_bf16.shape) == tuple(self.cls_score.weight.shape)
        self._bank_bf16 = bank_bf16
        self._bank_bf16_key = self._bank_key()

    def _cls_bias_is_zero(self) -> bool:
        b = self.cls_score.bias
        if b is None:
            return True
        key = (b.data_ptr(), b._version)
        if getattr(self, "_bias_zero_key", None) != key:
            self._bias_zero = not bool(b.detach().any())        # one host read per (re-)assignment of the bias
            self._bias_zero_key = key
        return self._bias_zero

    # ------------------------------------------------------------------ bank install (:214-236)
    def set_class_embeddings(self, embs):
        device = self.emb_pred.weight.device
        self.num_classes = embs.shape[0] - 1          # includes background
        self.cls_score = nn.Linear(self.emb_dim, self.num_classes + 1)
        self.cls_score.to(device)
        if torch.is_tensor(embs):
            embs = embs.clone().detach().to(device)
        else:
            embs = torch.tensor(embs, device=device)
        embs = embs.to(torch.float32)
        if self.normalize_emb:
            assert embs.shape[1] == self.emb_dim, "The embedding dimension has to match the one saved in the model"
            embs = _rownorm(embs, ops.NORM_L2)
        if self.standardize_emb:
            assert embs.shape[1] == self.emb_dim, "The embedding dimension has to match the one saved in the model"
            embs = _rownorm(embs, ops.NORM_STANDARDIZE)
        self.cls_score.weight.data = embs.contiguous()
        self.cls_score.bias.data = torch.zeros_like(self.cls_score.bias.data)
        self.cls_score.weight.requires_grad = False
        self.cls_score.bias.requires_grad = False
        self._bank_bf16 = None
        self._bank_bf16_key = None
        self._split_cache.pop("bank", None)           # the packed (hi, lo) copy belongs to the bank that was just replaced
        self._bias_zero_key = None


def _rownorm(x: torch.Tensor, mode: int) -> torch.Tensor:
    """normalize_vec / standardize_vec (logged_module.py:55-72) on the HIP kernels, forward and -- when the class predictor
    is trained through them -- backward (locov_rownorm_fwd / _bwd)."""
    return ops.rownorm_autograd(x, mode)


def build_box_predictor(cfg, input_shape):
    """box_emb_head.py:239-249."""
    name = cfg.MODEL.ROI_BOX_HEAD.NAME
    predictors = {
        "FastRCNNOutputLayers": FastRCNNOutputLayers,
        "EmbeddingFastRCNNOutputLayers": EmbeddingFastRCNNOutputLayers,
    }
    if name == "EmbeddingGroundingFastRCNNOutputLayers":      # SURVEY.md 8f-4 (imported late: it imports this module)
        from .box_emb_grounding_head import EmbeddingGroundingFastRCNNOutputLayers
        predictors[name] = EmbeddingGroundingFastRCNNOutputLayers
    if name not in predictors:
        raise KeyError(f"box predictor {name!r} is not part of the LSM ROI-head path")
    return predictors[name](cfg, input_shape)
