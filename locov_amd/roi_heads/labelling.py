"""Proposal labelling / sampling of the LSM ROI heads' training forward -- the [D2-upstream] ROIHeads base pieces (Matcher,
subsample_labels, add_ground_truth_to_proposals, ROIHeads._sample_proposals) and SampleAllROIHeads.label_and_sample_proposals of
ovr/modeling/roi_heads/roi_emb_heads.py:25-118, split into a device half that waits for nothing (_label_begin), a speculated
sample (_label_speculate / _label_validate: locov_label_proposals + locov_sample_proposals) and the host-driven form behind ONE
host read per batch (_label_finish).  The ROI heads themselves are in roi_emb_heads.py (which re-exports every name here).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch
from torch import nn

from .. import ops
from ..poolers import convert_boxes_to_pooler_format
from ..structures import Boxes, Instances, pairwise_iou

_SPECULATE = os.environ.get("LOCOV_LABEL_SPECULATE", "1") != "0"      # developer A/B: 0 = the training forwards wait for the labelling
_SAMPLE_KERNEL = os.environ.get("LOCOV_LABEL_SAMPLE_KERNEL", "1") != "0"  # developer A/B: 0 = the speculated sample from torch ops (two sorts, ~35 launches)

__all__ = ["get_event_storage", "Matcher", "subsample_labels", "subsample_order", "add_ground_truth_to_proposals", "ROIHeads",
           "SampleAllROIHeads"]


class _Events:
    """Scalar sink standing in for detectron2.utils.events.get_event_storage()."""

    def __init__(self):
        self.scalars: Dict[str, float] = {}

    def put_scalar(self, name, value):
        self.scalars[name] = float(value)


_EVENTS = _Events()


def get_event_storage():
    try:    # use Detectron2's storage when training under its trainer
        from detectron2.utils.events import get_event_storage as _g
        return _g()
    except Exception:
        return _EVENTS


class Matcher:
    """[D2-upstream] Matcher: per-prediction best GT and a label from IoU thresholds."""

    def __init__(self, thresholds: List[float], labels: List[int], allow_low_quality_matches: bool = False):
        thresholds = list(thresholds)
        assert thresholds[0] > 0
        thresholds.insert(0, -float("inf"))
        thresholds.append(float("inf"))
        assert all(low <= high for (low, high) in zip(thresholds[:-1], thresholds[1:]))
        assert all(l in [-1, 0, 1] for l in labels)
        assert len(labels) == len(thresholds) - 1
        self.thresholds, self.labels = thresholds, labels
        self.allow_low_quality_matches = allow_low_quality_matches
        self.check_quality = True          # host-side assert of Detectron2's Matcher (SampleAllROIHeads folds it into its one read)

    def __call__(self, match_quality_matrix: torch.Tensor):
        assert match_quality_matrix.dim() == 2
        if match_quality_matrix.numel() == 0:
            default_matches = match_quality_matrix.new_full((match_quality_matrix.size(1),), 0, dtype=torch.int64)
            default_match_labels = match_quality_matrix.new_full((match_quality_matrix.size(1),), self.labels[0],
                                                                 dtype=torch.int8)
            return default_matches, default_match_labels
        # (Detectron2 asserts `torch.all(match_quality_matrix >= 0)` here, a host sync per image; the labelling below reads the
        # same bit for the whole batch with its ONE host read and raises the AssertionError there -- `check_quality` is for
        # other callers)
        if self.check_quality:
            assert torch.all(match_quality_matrix >= 0)
        matched_vals, matches = match_quality_matrix.max(dim=0)
        match_labels = matches.new_full(matches.size(), 1, dtype=torch.int8)
        for (l, low, high) in zip(self.labels, self.thresholds[:-1], self.thresholds[1:]):
            low_high = (matched_vals >= low) & (matched_vals < high)
            match_labels[low_high] = l
        if self.allow_low_quality_matches:
            highest, _ = match_quality_matrix.max(dim=1)
            _, pred_inds = torch.nonzero(match_quality_matrix == highest[:, None], as_tuple=True)
            match_labels[pred_inds] = 1
        return matches, match_labels


def subsample_labels(labels: torch.Tensor, num_samples: int, positive_fraction: float, bg_label: int):
    """[D2-upstream] subsample_labels: random fg/bg subset (torch.randperm)."""
    positive = torch.nonzero((labels != -1) & (labels != bg_label), as_tuple=True)[0]
    negative = torch.nonzero(labels == bg_label, as_tuple=True)[0]
    num_pos = int(num_samples * positive_fraction)
    num_pos = min(positive.numel(), num_pos)
    num_neg = num_samples - num_pos
    num_neg = min(negative.numel(), num_neg)
    perm1 = torch.randperm(positive.numel(), device=positive.device)[:num_pos]
    perm2 = torch.randperm(negative.numel(), device=negative.device)[:num_neg]
    return positive[perm1], negative[perm2]


def subsample_order(labels: torch.Tensor, bg_label: int):
    """The device-only half of subsample_labels: a uniformly random order of all candidates with the foreground ones first,
    one with the background ones first, and the two population sizes as a device tensor.  Taking the first num_pos / num_neg
    entries (host integers, known once the sizes have been read -- ONE read for a whole batch) yields the same distribution
    as the randperm form, without a `nonzero` (= a host sync) per image."""
    pos = (labels != -1) & (labels != bg_label)
    neg = labels == bg_label
    k = torch.rand((2, labels.shape[0]), device=labels.device)
    pos_order = torch.argsort(k[0] + (~pos).to(k.dtype) * 2.0)
    neg_order = torch.argsort(k[1] + (~neg).to(k.dtype) * 2.0)
    return pos_order, neg_order, torch.stack([pos.sum(), neg.sum()])


def add_ground_truth_to_proposals(targets: List[Instances], proposals: List[Instances]) -> List[Instances]:
    """[D2-upstream] append the GT boxes to the proposals (objectness = logit(1 - 1e-10))."""
    assert len(proposals) == len(targets)
    if len(proposals) == 0:
        return proposals
    out = []
    for gt_i, prop_i in zip(targets, proposals):
        # (the caller's own container classes throughout: Detectron2's Instances / Boxes under train_ovnet.py)
        inst_cls = type(prop_i)
        gt_boxes = gt_i.gt_boxes if hasattr(gt_i, "get_fields") else gt_i
        device = prop_i.objectness_logits.device if prop_i.has("objectness_logits") else gt_boxes.device
        gt_logit_value = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))
        gt_proposal = inst_cls(prop_i.image_size)
        gt_proposal.proposal_boxes = gt_boxes
        if prop_i.has("objectness_logits"):
            gt_proposal.objectness_logits = torch.full((len(gt_boxes),), gt_logit_value, device=device)   # (= value * ones, one launch)
        keep = inst_cls(prop_i.image_size)
        for k in gt_proposal.get_fields():
            keep.set(k, prop_i.get(k))
        out.append(inst_cls.cat([keep, gt_proposal]))
    return out


class ROIHeads(nn.Module):
    """[D2-upstream] ROIHeads base: sampling hyper-parameters + proposal matcher."""

    def __init__(self, *, num_classes, batch_size_per_image, positive_fraction, proposal_matcher,
                 proposal_append_gt=True):
        super().__init__()
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.num_classes = num_classes
        self.proposal_matcher = proposal_matcher
        self.proposal_append_gt = proposal_append_gt

    @classmethod
    def from_config(cls, cfg):
        return {
            "batch_size_per_image": cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE,
            "positive_fraction": cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION,
            "num_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES,
            "proposal_append_gt": cfg.MODEL.ROI_HEADS.PROPOSAL_APPEND_GT,
            "proposal_matcher": Matcher(cfg.MODEL.ROI_HEADS.IOU_THRESHOLDS, cfg.MODEL.ROI_HEADS.IOU_LABELS,
                                        allow_low_quality_matches=False),
        }

    def _sample_proposals(self, matched_idxs, matched_labels, gt_classes):
        has_gt = gt_classes.numel() > 0
        if has_gt:
            gt_classes = gt_classes[matched_idxs]
            gt_classes[matched_labels == 0] = self.num_classes
            gt_classes[matched_labels == -1] = -1
        else:
            gt_classes = torch.zeros_like(matched_idxs) + self.num_classes
        sampled_fg_idxs, sampled_bg_idxs = subsample_labels(gt_classes, self.batch_size_per_image,
                                                            self.positive_fraction, self.num_classes)
        sampled_idxs = torch.cat([sampled_fg_idxs, sampled_bg_idxs], dim=0)
        return sampled_idxs, gt_classes[sampled_idxs]


class SampleAllROIHeads(ROIHeads):
    """Proposal labelling / sampling of ovr/modeling/roi_heads/roi_emb_heads.py:25-118.

    Kept from the reference (and different from stock Detectron2): EVERY field of the matched target
    is copied onto the sampled proposals, not only the gt_* ones (:97-100), and a 0/1 `fg_proposal`
    field is attached (:102-104).  Changed in HOW: the per-image fg/bg counters stay on the device and
    are read back once per call instead of two `.item()` host syncs per image (:109-110) -- and so are the sizes of the foreground /
    background populations the sampler needs (`subsample_order`): labelling a batch costs ONE host sync in total."""

    def _match_one_image(self, props: Instances, tgt: Instances):
        """Everything of one image's labelling that needs no host value: IoU, matching, class labels, sampling orders, and
        the two validity bits the reference asserts on the host (Matcher: IoU >= 0; Box2BoxTransform.get_deltas: every
        foreground proposal has positive width and height) as device values for the batch's one read."""
        iou = pairwise_iou(tgt.gt_boxes, props.proposal_boxes)                 # [num_gt, num_proposals]
        matcher = self.proposal_matcher
        was, matcher.check_quality = getattr(matcher, "check_quality", True), False
        try:
            gt_index, match_label = matcher(iou)
        finally:
            matcher.check_quality = was
        if tgt.gt_classes.numel() > 0:                                         # ROIHeads._sample_proposals' labelling
            labels = tgt.gt_classes[gt_index]
            labels[match_label == 0] = self.num_classes
            labels[match_label == -1] = -1
        else:
            labels = torch.zeros_like(gt_index) + self.num_classes
        pos_order, neg_order, counts = subsample_order(labels, self.num_classes)
        box = props.proposal_boxes.tensor
        fg = (labels >= 0) & (labels != self.num_classes)
        degenerate = ~(((box[:, 2] - box[:, 0]) > 0) & ((box[:, 3] - box[:, 1]) > 0)) & fg        # (NaN counts as invalid, as upstream)
        bad = torch.stack([~(iou >= 0).all() if iou.numel() else degenerate.new_zeros(()), degenerate.any()]).to(counts.dtype)
        return gt_index, labels, pos_order, neg_order, torch.cat([counts, bad])

    def _finish_one_image(self, props: Instances, tgt: Instances, gt_index, labels, pos_order, neg_order, n_pos_avail, n_neg_avail):
        num_pos = min(n_pos_avail, int(self.batch_size_per_image * self.positive_fraction))      # subsample_labels' counts
        num_neg = min(n_neg_avail, self.batch_size_per_image - num_pos)
        picked = torch.cat([pos_order[:num_pos], neg_order[:num_neg]], dim=0)
        classes = labels[picked]
        out = props[picked]
        out.gt_classes = classes
        if len(tgt) > 0:
            src = gt_index[picked]
            for name, value in tgt.get_fields().items():
                if not out.has(name):
                    out.set(name, value[src])
        is_bg = classes == self.num_classes
        out.set("fg_proposal", (~is_bg).to(classes.dtype))
        return out, num_neg, num_pos + num_neg

    @property
    def stats(self) -> Dict[str, int]:
        """Counters of the training forward's retry machine since construction (or `stats.clear()`): `forwards`; `speculated`
        (sample formed on the device without a host wait); `speculation_misses` (a speculated forward repeated from the true counts);
        `unspeculated` (the forward waited for the labelling: a batch the lean path does not take, e.g. an image with fewer
        candidates than the budget); `guard_trips` (a forward left the split arithmetic's range) and `fp32_repeats` (RES5_TRAIN_GUARD
        "sync": that forward repeated on the f32 MFMA); `deferred_trips` ("deferred": a skipped step found at the next labelling read);
        `bwd_guard_trips` (a stale backward weight scale: that backward's Res5 gradients were zeroed)."""
        return self.__dict__.setdefault("_stats", {})

    def _count(self, key: str, n: int = 1) -> None:
        st = self.stats
        st[key] = st.get(key, 0) + n

    def _backward_guard_tripped(self) -> None:
        pass

    def _deferred_guards(self, device):
        """[(kind, RangeGuard)] whose words travel with the labelling read (none in the base class)."""
        return []

    def _deferred_guards_tripped(self, kinds) -> None:
        pass

    def _label_and_sample_per_image(self, proposals: List[Instances], targets: List[Instances]) -> List[Instances]:
        """The per-image form of label_and_sample_proposals (any matcher object: one that asserts a non-negative quality matrix
        cannot take the batch form's -1 mask).  Same results, ~4x the launches."""
        sampled, bg_counts, totals = [], [], []
        matched = [self._match_one_image(props, tgt) for props, tgt in zip(proposals, targets)]     # no host value needed
        # ONE host read for the whole batch: population sizes, the reference's two validity asserts, and -- it costs nothing
        # here -- the range-guard words the previous step's Res5 backward may have raised (res5_train.Res5BlockFn.backward)
        avail = []
        if matched:
            rows = torch.stack([m[-1] for m in matched])
            guards = self._deferred_guards(rows.device)
            if guards:
                flat = torch.cat([rows.reshape(-1), torch.stack([g.word.reshape(()) for _, g in guards]).to(rows.dtype)]).cpu()
                rows_h, guard_h = flat[:rows.numel()].view(rows.shape), flat[rows.numel():]
                tripped = {kind for (kind, _), v in zip(guards, guard_h.tolist()) if v}
                if tripped:
                    self._deferred_guards_tripped(tripped)
            else:
                rows_h = rows.cpu()
            assert not bool(rows_h[:, 2].any()), "Matcher: the match quality matrix has negative entries"
            assert not bool(rows_h[:, 3].any()), "Input boxes to Box2BoxTransform are not valid!"
            avail = rows_h[:, :2].tolist()
        for props, tgt, m, (n_pos_avail, n_neg_avail) in zip(proposals, targets, matched, avail):
            out, n_bg, n = self._finish_one_image(props, tgt, *m[:-1], int(n_pos_avail), int(n_neg_avail))
            sampled.append(out)
            bg_counts.append(n_bg)
            totals.append(n)
        if sampled:
            bg = np.asarray(bg_counts, dtype=np.float64)
            tot = np.asarray(totals, dtype=np.float64)
            storage = get_event_storage()
            storage.put_scalar("roi_head/num_fg_samples", float(np.mean(tot - bg)))
            storage.put_scalar("roi_head/num_bg_samples", float(np.mean(bg)))
        return sampled


    def _match_batch(self, proposals: List[Instances], targets: List[Instances]):
        """The device half of labelling a whole BATCH in one set of launches (what _match_one_image does per image: the batch of
        a training step is 4 images x ~60 small launches -- a third of the step's launch count): IoU of every ground-truth box
        with every proposal of the batch, pairs from different images masked to -1 so that they can never be the maximum
        (the same argmax, hence the same matches and labels, as the per-image form), the reference's class labels, ONE pair
        of sorts for the sampling orders (key = image * 4 + [not in the population] * 2 + U[0,1): image-major, population
        first, uniformly random inside it), per-image population sizes and the two validity bits.
        Returns (gt_index [global], labels, pos_order, neg_order, rows [B,4] = n_pos, n_neg, bad IoU, degenerate fg box)."""
        dev = proposals[0].proposal_boxes.tensor.device
        n_r = [len(p) for p in proposals]
        n_g = [len(t) for t in targets]
        B = len(proposals)
        box = torch.cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        matcher = self.proposal_matcher
        if (box.is_cuda and box.dtype == torch.float32 and 0 < B <= ops.LABEL_MAX_IMAGES and len(matcher.labels) <= ops.LABEL_MAX_THRESHOLDS
                and not matcher.allow_low_quality_matches and box.shape[0] > 0):
            # ONE kernel for IoU / matching / labels / sort keys / counts (locov_label_proposals: the torch ops below, bit for bit,
            # without the [sum M, sum R] matrix and ~80 small launches), then the two sorts
            if sum(n_g) > 0:
                gtb = torch.cat([t.gt_boxes.tensor for t in targets], dim=0).to(torch.float32)
                gtc = torch.cat([t.gt_classes for t in targets], dim=0).to(torch.int64)
            else:
                gtb = gtc = None
            k = torch.rand((2, box.shape[0]), device=dev, dtype=torch.float64)
            gt_index, labels, key_pos, key_neg, rows = ops.label_proposals(box, n_r, gtb, gtc, n_g, matcher.thresholds, matcher.labels,
                                                                           self.num_classes, k)
            return gt_index, labels, torch.argsort(key_pos), torch.argsort(key_neg), rows
        img_r = torch.cat([torch.full((n,), i, dtype=torch.int64, device=dev) for i, n in enumerate(n_r)])
        counts_dtype = torch.int64
        if sum(n_g) > 0:
            gtb = torch.cat([t.gt_boxes.tensor for t in targets], dim=0)
            gtc = torch.cat([t.gt_classes for t in targets], dim=0)
            img_g = torch.cat([torch.full((n,), i, dtype=torch.int64, device=dev) for i, n in enumerate(n_g)])
            iou = pairwise_iou(Boxes(gtb), Boxes(box))                         # [sum M, sum R]
            same = img_g[:, None] == img_r[None, :]
            bad_iou = (~(iou >= 0) & same).any()
            quality = torch.where(same, iou, torch.full((), -1.0, dtype=iou.dtype, device=dev))
            matcher = self.proposal_matcher
            was, matcher.check_quality = getattr(matcher, "check_quality", True), False
            try:
                gt_index, match_label = matcher(quality)                       # gt_index: row of the CONCATENATED targets
            finally:
                matcher.check_quality = was
            labels = gtc[gt_index]                                              # ROIHeads._sample_proposals' labelling
            labels[match_label == 0] = self.num_classes
            labels[match_label == -1] = -1
            if min(n_g) == 0:
                # an image without ground truth inside a batch that has some: _sample_proposals' has_gt == False branch labels all
                # of its proposals background, whatever the Matcher's "no match" label is (IOU_LABELS[0] may be -1)
                no_gt = torch.tensor([n == 0 for n in n_g], device=dev)[img_r]
                labels[no_gt] = self.num_classes
        else:
            gt_index = torch.zeros(box.shape[0], dtype=torch.int64, device=dev)
            labels = torch.zeros_like(gt_index) + self.num_classes
            bad_iou = torch.zeros((), dtype=torch.bool, device=dev)
        pos = (labels != -1) & (labels != self.num_classes)
        neg = labels == self.num_classes
        k = torch.rand((2, box.shape[0]), device=dev, dtype=torch.float64)      # (float64: the image term must not cost the draw its bits)
        base = img_r.to(k.dtype) * 4.0
        pos_order = torch.argsort(k[0] + (~pos).to(k.dtype) * 2.0 + base)
        neg_order = torch.argsort(k[1] + (~neg).to(k.dtype) * 2.0 + base)
        degenerate = ~(((box[:, 2] - box[:, 0]) > 0) & ((box[:, 3] - box[:, 1]) > 0)) & pos      # (NaN counts as invalid, as upstream)
        per_image = torch.zeros((3, B), dtype=counts_dtype, device=dev)
        per_image.index_add_(1, img_r, torch.stack([pos, neg, degenerate]).to(counts_dtype))
        rows = torch.cat([per_image[:2], bad_iou.to(counts_dtype).expand(1, B), per_image[2:]], dim=0).t().contiguous()
        return gt_index, labels, pos_order, neg_order, rows

    @staticmethod
    def _gather_split(values, index: torch.Tensor, sizes: List[int]):
        """values: one field of every image (tensors or Boxes-like objects with `.tensor`) -> that field of the sampled
        instances: ONE concatenation and ONE gather for the batch, handed out as per-image views."""
        v0 = values[0]
        boxes_like = not isinstance(v0, torch.Tensor) and hasattr(v0, "tensor")
        if not boxes_like and not isinstance(v0, torch.Tensor):
            # any other indexable field type (lists, masks, ...): index image by image with the image's own row numbers
            offs = np.concatenate([[0], np.cumsum([len(v) for v in values])]).tolist()
            return [v[idx - off] for v, idx, off in zip(values, torch.split(index, sizes), offs)]
        flat = torch.cat([v.tensor if boxes_like else v for v in values], dim=0)[index]
        parts = torch.split(flat, sizes, dim=0)
        return [type(v0)(p) for p in parts] if boxes_like else list(parts)

    @torch.no_grad()
    def label_and_sample_proposals(self, proposals: List[Instances], targets: List[Instances]) -> List[Instances]:
        return self._label_finish(self._label_begin(proposals, targets))

    @torch.no_grad()
    def _label_begin(self, proposals: List[Instances], targets: List[Instances]):
        """The device half of label_and_sample_proposals: everything is enqueued, the few integers the host needs are on their
        way to pinned memory behind an event -- and NOTHING waits.  A caller with independent device work (the whole-grid Res5
        call of EmbeddingProposalsRes5ROIHeads.forward) enqueues it between _label_begin and _label_finish: the host then waits
        for the labelling kernels only, with that work still queued behind them, instead of draining the GPU once per step."""
        lean = self._label_lean_inputs(proposals, targets)
        if lean is not None:
            # the training step's batch (device fp32 boxes, the stock Matcher, every image at least its budget of candidates):
            # one concatenation per field, the labelling kernel, and -- _label_speculate -- the sampling kernel; the per-image
            # Instances with the ground truth appended and the two sorts of the host-driven form are only built if it is needed
            gt_index, labels, key_pos, key_neg, rows = ops.label_proposals(lean["box"], lean["n_r"], lean["gtb"], lean["gtc"], lean["n_g"],
                                                                           self.proposal_matcher.thresholds, self.proposal_matcher.labels,
                                                                           self.num_classes, lean["rnd"])
            raw, pos_order, neg_order = proposals, None, None
            proposals = None
        else:
            raw, key_pos, key_neg = None, None, None
            if self.proposal_append_gt:
                proposals = add_ground_truth_to_proposals(targets, proposals)
            if not proposals:
                return {"done": []}
            if type(self.proposal_matcher) is not Matcher:
                return {"done": self._label_and_sample_per_image(proposals, targets)}
            gt_index, labels, pos_order, neg_order, rows = self._match_batch(proposals, targets)        # no host value needed
        # ONE host read for the whole batch: population sizes, the reference's two validity asserts, and -- it costs nothing
        # here -- the deferred range-guard words of the previous step (Res5's backward, the training forward)
        guards = self._deferred_guards(rows.device)
        flat = rows.reshape(-1)
        event, host_g = None, None
        if flat.is_cuda:
            # (one small copy per tensor straight into pinned memory: no concatenation / dtype-conversion launches in front of it)
            host = self.__dict__.get("_label_pinned")
            if host is None or host.numel() < flat.numel() or host.dtype != flat.dtype:
                host = self.__dict__["_label_pinned"] = torch.empty(max(flat.numel(), 256), dtype=flat.dtype).pin_memory()
            host = host[:flat.numel()]
            host.copy_(flat, non_blocking=True)
            if guards:
                host_g = self.__dict__.get("_label_pinned_guards")
                if host_g is None or host_g.numel() < len(guards) or host_g.dtype != guards[0][1].word.dtype:
                    host_g = self.__dict__["_label_pinned_guards"] = torch.empty(max(len(guards), 8), dtype=guards[0][1].word.dtype).pin_memory()
                for i, (_, g) in enumerate(guards):
                    host_g[i:i + 1].copy_(g.word.reshape(1), non_blocking=True)
            event = torch.cuda.Event()
            event.record(torch.cuda.current_stream(flat.device))
        else:
            host = flat
            if guards:
                host_g = torch.stack([g.word.reshape(()) for _, g in guards]).cpu()
        st = {"targets": targets, "gt_index": gt_index, "labels": labels, "rows": rows, "rows_shape": tuple(rows.shape), "host": host,
              "host_guards": host_g, "event": event, "guards": guards}
        if lean is not None:
            st.update(lean=lean, raw_proposals=raw, keys=(key_pos, key_neg))
        else:
            st.update(proposals=proposals, pos_order=pos_order, neg_order=neg_order)
        return st

    def _label_lean_inputs(self, proposals: List[Instances], targets: List[Instances]):
        """The concatenated inputs of the one-launch labelling + one-launch sampling, or None when the batch is not of the
        training step's plain kind: proposals carrying proposal_boxes (+ objectness_logits), targets carrying gt_boxes + gt_classes,
        fp32 device boxes, the stock Matcher without low-quality matches, and per image (ground truth appended) between the
        budget and ops.SAMPLE_MAX_PROPOSALS candidates."""
        if not (_SPECULATE and _SAMPLE_KERNEL) or not proposals or len(proposals) != len(targets) or len(proposals) > ops.LABEL_MAX_IMAGES:
            return None
        matcher = self.proposal_matcher
        if type(matcher) is not Matcher or matcher.allow_low_quality_matches or len(matcher.labels) > ops.LABEL_MAX_THRESHOLDS:
            return None
        with_logits = proposals[0].has("objectness_logits")
        fields = {"proposal_boxes", "objectness_logits"} if with_logits else {"proposal_boxes"}
        B = int(self.batch_size_per_image)
        append = bool(self.proposal_append_gt)
        pieces, logit_pieces, gtbs, gtcs, n_r, n_g = [], [], [], [], [], []
        for p, t in zip(proposals, targets):
            if set(p.get_fields()) != fields or set(t.get_fields()) != {"gt_boxes", "gt_classes"} or type(t) is not type(p):
                return None
            pb, gb, gc = p.proposal_boxes.tensor, t.gt_boxes.tensor, t.gt_classes
            if not (pb.is_cuda and pb.dtype == torch.float32 and gb.dtype == torch.float32 and gb.device == pb.device
                    and gc.dtype == torch.int64 and gc.device == pb.device and type(t.gt_boxes) is type(p.proposal_boxes)):
                return None
            n = len(p) + (len(t) if append else 0)
            if not (max(B, 1) <= n <= ops.SAMPLE_MAX_PROPOSALS) or B <= 0:
                return None
            pieces.append(pb)
            if with_logits:
                lg = p.objectness_logits
                if not (lg.dtype == torch.float32 and lg.dim() == 1 and lg.device == pb.device):
                    return None
                logit_pieces.append(lg)
            if append and len(t):
                pieces.append(gb)
                if with_logits:
                    logit_pieces.append(self._gt_logits(len(t), pb.device))
            if len(t):
                gtbs.append(gb)
                gtcs.append(gc)
            n_r.append(n)
            n_g.append(len(t))
        dev = pieces[0].device
        box = torch.cat(pieces, dim=0)
        return {"box": box, "n_r": n_r, "n_g": n_g, "gtb": torch.cat(gtbs, dim=0) if gtbs else None,
                "gtc": torch.cat(gtcs, dim=0) if gtcs else None, "logits": torch.cat(logit_pieces, dim=0) if with_logits else None,
                "rnd": torch.rand((2, box.shape[0]), device=dev, dtype=torch.float64)}

    def _gt_logits(self, n: int, device) -> torch.Tensor:
        """n objectness logits of appended ground-truth boxes (add_ground_truth_to_proposals' constant), a view of a cached tensor."""
        hit = self.__dict__.get("_gt_logit_const")
        if hit is None or hit.numel() < n or hit.device != device:
            hit = self.__dict__["_gt_logit_const"] = torch.full((max(n, 64),), math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10))), device=device)
        return hit[:n]

    def _label_materialize(self, st) -> None:
        """What the host-driven sampling needs and the lean labelling skipped: the per-image candidates with the ground truth
        appended, and the two global sampling orders."""
        if "proposals" not in st:
            raw = st["raw_proposals"]
            st["proposals"] = add_ground_truth_to_proposals(st["targets"], raw) if self.proposal_append_gt else raw
            st["pos_order"], st["neg_order"] = torch.argsort(st["keys"][0]), torch.argsort(st["keys"][1])

    @torch.no_grad()
    def _label_host(self, st):
        """The host half of the labelling's ONE read: wait for the event behind the labelling kernels, act on the deferred
        range-guard words that travelled with it, raise the reference's two asserts, and return the per-image population sizes
        [(foreground candidates, background candidates)].  Idempotent per labelling."""
        if "avail" in st:
            return st["avail"]
        if st["event"] is not None:
            st["event"].synchronize()                       # the step's host wait for the labelling kernels
        n_rows = st["rows_shape"][0] * st["rows_shape"][1]
        flat = st["host"].clone() if st["event"] is not None else st["host"].cpu()
        rows_h = flat[:n_rows].view(st["rows_shape"])
        guard_h = st["host_guards"][:len(st["guards"])].tolist() if st["guards"] else []
        tripped = {kind for (kind, _), v in zip(st["guards"], guard_h) if v}
        if tripped:
            self._deferred_guards_tripped(tripped)
        assert not bool(rows_h[:, 2].any()), "Matcher: the match quality matrix has negative entries"
        assert not bool(rows_h[:, 3].any()), "Input boxes to Box2BoxTransform are not valid!"
        st["avail"] = rows_h[:, :2].tolist()
        return st["avail"]

    def _sample_counts(self, avail):
        """subsample_labels' counts per image: [(num_pos, num_neg)]."""
        out = []
        for n_pos_avail, n_neg_avail in avail:
            num_pos = min(int(n_pos_avail), int(self.batch_size_per_image * self.positive_fraction))
            out.append((num_pos, min(int(n_neg_avail), self.batch_size_per_image - num_pos)))
        return out

    @torch.no_grad()
    def _label_build(self, st, picked: torch.Tensor, sizes: List[int], speculated: bool = False) -> List[Instances]:
        """The sampled Instances of a batch from `picked` (rows of the concatenated candidates, image-major) and the per-image
        sample counts: ONE gather per field for the batch (the sampled rows of image i are a slice of the two image-major orders:
        its foreground draws sit at the head of its segment of pos_order, likewise neg_order)."""
        proposals, targets = st["proposals"], st["targets"]
        gt_index, labels = st["gt_index"], st["labels"]
        n_g = [len(t) for t in targets]
        off_g = np.concatenate([[0], np.cumsum(n_g)]).tolist()
        picked_labels = labels[picked]
        if speculated:
            # (a speculated sample of a batch that does NOT fill its budget reads past the populations: such rows may carry the
            # ignore label -1, which the cross-entropy enqueued on this sample answers with a device-side assert before the
            # validation can throw the sample away -- they count as background)
            picked_labels = torch.where(picked_labels < 0, torch.full_like(picked_labels, self.num_classes), picked_labels)
        classes = torch.split(picked_labels, sizes)
        fg_flags = torch.split((picked_labels != self.num_classes).to(picked_labels.dtype), sizes)     # (one launch pair for the batch)
        src_global = gt_index[picked]
        prop_fields = list(proposals[0].get_fields().keys())
        sampled = [type(p)(p.image_size) for p in proposals]
        for name in prop_fields:
            for out, part in zip(sampled, self._gather_split([p.get(name) for p in proposals], picked, sizes)):
                out.set(name, part)
        for out, cls in zip(sampled, classes):
            out.gt_classes = cls
        # every field of the matched target (:97-100) -- images without ground truth carry none
        with_gt = [i for i in range(len(targets)) if n_g[i] > 0]
        if with_gt:
            tgt_fields = [k for k in targets[with_gt[0]].get_fields() if not sampled[with_gt[0]].has(k)]
            if len(with_gt) == len(targets):
                for name in tgt_fields:
                    for out, part in zip(sampled, self._gather_split([t.get(name) for t in targets], src_global, sizes)):
                        out.set(name, part)
            else:
                src_parts = torch.split(src_global, sizes)
                for i in with_gt:
                    src = src_parts[i] - off_g[i]
                    for name in tgt_fields:
                        sampled[i].set(name, targets[i].get(name)[src])
        for out, flag in zip(sampled, fg_flags):
            out.set("fg_proposal", flag)
        return sampled

    def _label_log(self, counts, st=None) -> None:
        """roi_head/num_{fg,bg}_samples (:114-116), ONCE per labelling: a forward that is repeated (the read flipped RES5_DTYPE, a
        speculation miss) must not enter the event storage's smoothed history twice."""
        if st is not None:
            if st.get("logged"):
                return
            st["logged"] = True
        bg = np.asarray([n for _, n in counts], dtype=np.float64)
        fg = np.asarray([p for p, _ in counts], dtype=np.float64)
        storage = get_event_storage()
        storage.put_scalar("roi_head/num_fg_samples", float(np.mean(fg)))
        storage.put_scalar("roi_head/num_bg_samples", float(np.mean(bg)))

    @torch.no_grad()
    def _label_finish(self, st) -> List[Instances]:
        if "done" in st:
            return st["done"]
        counts = self._sample_counts(self._label_host(st))
        self._label_materialize(st)
        pos_order, neg_order = st["pos_order"], st["neg_order"]
        off_r = np.concatenate([[0], np.cumsum([len(p) for p in st["proposals"]])]).tolist()
        pieces = []
        for i, (num_pos, num_neg) in enumerate(counts):
            pieces += [pos_order[off_r[i]:off_r[i] + num_pos], neg_order[off_r[i]:off_r[i] + num_neg]]
        picked = torch.cat(pieces, dim=0)                                       # rows of the concatenated proposals
        sampled = self._label_build(st, picked, [p + n for p, n in counts])
        self._label_log(counts, st)
        st["done"] = sampled                              # (a forward that is repeated on the f32 MFMA keeps its draw)
        return sampled

    @torch.no_grad()
    def _label_speculate(self, st) -> Optional[List[Instances]]:
        """The sampled Instances WITHOUT the host read, on the assumption that every image fills its budget of
        batch_size_per_image samples (it does whenever it has batch_size_per_image - num_pos background candidates: 1 000
        proposals against a budget of 200 / 512): the per-image counts then only decide where the foreground draws end inside
        each image's rows, which a device select can do.  The caller enqueues everything that depends on the sample (ROIAlign,
        Res5, predictor, losses) and validates afterwards (_label_validate) at a point where it waits for the GPU anyway; a
        batch that does not fill its budget is redone from _label_finish.  None when the assumption cannot hold or nothing
        would be gained (host tensors, a foreign matcher, fewer candidates than the budget)."""
        if "done" in st or st.get("event") is None or not _SPECULATE:
            return None
        B = int(self.batch_size_per_image)
        lean = st.get("lean")
        if lean is not None:
            # labels, keys and counts are on the device: ONE launch sorts every image's candidates by the two keys, takes the budget
            # and gathers every field of the sampled Instances (and the pooler's rois)
            picked, boxes, classes, gt_boxes, fg, rois, logits = ops.sample_proposals(
                st["keys"][0], st["keys"][1], st["labels"], st["gt_index"], st["rows"], lean["box"], lean["gtb"], lean["n_r"], lean["n_g"],
                B, int(B * self.positive_fraction), self.num_classes, field=lean["logits"])
            sampled = []
            for i, (p, t) in enumerate(zip(st["raw_proposals"], st["targets"])):
                sl = slice(i * B, (i + 1) * B)
                out = type(p)(p.image_size)
                out.set("proposal_boxes", type(p.proposal_boxes)(boxes[sl]))
                if logits is not None:
                    out.set("objectness_logits", logits[sl])
                out.gt_classes = classes[sl]
                if lean["n_g"][i] > 0:
                    out.set("gt_boxes", type(t.gt_boxes)(gt_boxes[sl]))
                out.set("fg_proposal", fg[sl])
                sampled.append(out)
            st.update(speculated=True, picked=picked, rois=rois, rois_of=sampled)
            return sampled
        self._label_materialize(st)
        n_r = [len(p) for p in st["proposals"]]
        if B <= 0 or min(n_r) < B:
            return None
        rows, pos_order, neg_order = st["rows"], st["pos_order"], st["neg_order"]
        dev = rows.device
        # (the per-image row offsets and the slot numbers only depend on the batch's shape: built once -- a host-to-device copy of
        # pageable memory per step would wait for everything the stream still holds)
        key = (tuple(n_r), B, dev)
        cached = self.__dict__.get("_spec_index")
        if cached is None or cached[0] != key:
            off = torch.tensor(np.concatenate([[0], np.cumsum(n_r)[:-1]]), dtype=torch.int64).to(dev)[:, None]   # [n_img, 1]
            cached = self.__dict__["_spec_index"] = (key, off, torch.arange(B, dtype=torch.int64, device=dev)[None, :])
        _, off, j = cached
        num_pos = rows[:, 0].clamp(max=int(B * self.positive_fraction))[:, None]                                    # [n_img, 1]
        last = pos_order.numel() - 1
        picked = torch.where(j < num_pos, pos_order[(off + j).clamp(max=last)], neg_order[(off + (j - num_pos).clamp(min=0)).clamp(max=last)])
        st["speculated"] = True
        return self._label_build(st, picked.reshape(-1), [B] * len(n_r), speculated=True)

    @staticmethod
    def _sampled_rois(st, sampled: List[Instances]) -> torch.Tensor:
        """The pooler's [R, 5] input of a sampled batch: the sampling kernel's own output when `sampled` is what it produced."""
        if st.get("rois_of") is sampled:
            return st["rois"]
        return convert_boxes_to_pooler_format([x.proposal_boxes for x in sampled])

    @torch.no_grad()
    def _label_validate(self, st) -> bool:
        """The host read behind a speculated sample: the reference's asserts, the deferred guard words, the logged counts -- and
        whether every image did fill its budget (else the caller repeats the step from _label_finish)."""
        counts = self._sample_counts(self._label_host(st))
        ok = all(p + n == int(self.batch_size_per_image) for p, n in counts)
        if ok:
            self._label_log(counts, st)
        return ok
