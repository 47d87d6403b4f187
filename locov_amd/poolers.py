"""ROIPooler with Detectron2's constructor and call signature, running on the gfx950 kernels.

[D2-upstream] detectron2.modeling.poolers.ROIPooler as the reference builds it
(ovr/modeling/roi_heads/roi_emb_heads.py:182-187: output_size=14, scales=(1/16,),
sampling_ratio=0, pooler_type="ROIAlignV2") and calls it (:244).  Differences in HOW, not
WHAT: multi-level pooling is one launch (level index per ROI, computed by
locov_level_assign) instead of a per-level nonzero / index_put_ loop with host syncs.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Union

import torch
from torch import nn

from . import ops
from .structures import Boxes, boxes_tensor

__all__ = ["ROIPooler", "convert_boxes_to_pooler_format", "assign_boxes_to_levels"]


_BATCH_INDEX_COLUMNS = {}


def _batch_index_column(lens, dev) -> torch.Tensor:
    """[R,1] fp32 column of image indices for per-image row counts `lens` (host-side shapes): built once per (device, counts) by
    fill kernels -- no host-to-device copy, no sync -- and re-used: a detector sees the same proposal counts call after call."""
    key = (dev, lens)
    col = _BATCH_INDEX_COLUMNS.get(key)
    if col is None:
        if len(_BATCH_INDEX_COLUMNS) >= 16:
            _BATCH_INDEX_COLUMNS.pop(next(iter(_BATCH_INDEX_COLUMNS)))
        col = torch.empty((sum(lens), 1), dtype=torch.float32, device=dev)
        r0 = 0
        for i, n in enumerate(lens):
            col[r0:r0 + n] = float(i)
            r0 += n
        _BATCH_INDEX_COLUMNS[key] = col
    return col


def convert_boxes_to_pooler_format(box_lists: Sequence[Union[Boxes, torch.Tensor]]) -> torch.Tensor:
    """per-image [Ri,4] boxes -> [R,5] rows (batch_index, x0, y0, x1, y1).  Two launches whatever the number of images (the boxes
    concatenated, then joined to the cached index column) instead of two per image."""
    tensors = [boxes_tensor(b) for b in box_lists]
    if len(tensors) == 0:
        return torch.zeros((0, 5), dtype=torch.float32)
    dev = tensors[0].device
    boxes = torch.cat(tensors, dim=0) if len(tensors) > 1 else tensors[0]
    if boxes.dtype != torch.float32:
        boxes = boxes.to(torch.float32)
    return torch.cat([_batch_index_column(tuple(int(t.shape[0]) for t in tensors), dev), boxes], dim=1)


def assign_boxes_to_levels(box_lists, min_level: int, max_level: int, canonical_box_size: int,
                           canonical_level: int) -> torch.Tensor:
    boxes = torch.cat([boxes_tensor(b) for b in box_lists], dim=0).to(torch.float32).contiguous()
    return ops.level_assign(boxes, min_level, max_level, canonical_box_size, canonical_level)


class ROIPooler(nn.Module):
    def __init__(self, output_size, scales, sampling_ratio, pooler_type, canonical_box_size=224,
                 canonical_level=4):
        super().__init__()
        if isinstance(output_size, int):
            output_size = (output_size, output_size)
        assert len(output_size) == 2 and isinstance(output_size[0], int) and isinstance(output_size[1], int)
        self.output_size = output_size
        if pooler_type == "ROIAlign":
            self.aligned = False
        elif pooler_type == "ROIAlignV2":
            self.aligned = True
        else:   # ROIPool / ROIAlignRotated are never selected by the reference's configs
            raise ValueError(f"Unsupported pooler type for the LSM ROI head: {pooler_type}")
        self.pooler_type = pooler_type
        self.scales = tuple(float(s) for s in scales)
        self.sampling_ratio = int(sampling_ratio)
        min_level = -(math.log2(self.scales[0]))
        max_level = -(math.log2(self.scales[-1]))
        assert math.isclose(min_level, int(min_level)) and math.isclose(max_level, int(max_level)), \
            "Featuremap stride is not power of 2!"
        self.min_level, self.max_level = int(min_level), int(max_level)
        assert len(self.scales) == self.max_level - self.min_level + 1, \
            "[ROIPooler] Sizes of input featuremaps do not form a pyramid!"
        assert 0 <= self.min_level <= self.max_level
        self.canonical_level = canonical_level
        assert canonical_box_size > 0
        self.canonical_box_size = canonical_box_size

    def forward(self, x: List[torch.Tensor], box_lists) -> torch.Tensor:
        num_levels = len(self.scales)
        assert isinstance(x, list) and isinstance(box_lists, list), "Arguments to pooler must be lists"
        assert len(x) == num_levels, f"unequal value, num_level_assignments={num_levels}, but x is list of {len(x)} Tensors"
        assert len(box_lists) == x[0].size(0), \
            f"unequal value, x[0] batch dim 0 is {x[0].size(0)}, but box_list has length {len(box_lists)}"
        if len(box_lists) == 0:
            return torch.zeros((0, x[0].shape[1]) + self.output_size, device=x[0].device, dtype=x[0].dtype)
        rois = convert_boxes_to_pooler_format(box_lists)
        if num_levels == 1:
            return ops.roi_align(x[0], rois, self.output_size, self.scales[0], self.sampling_ratio, self.aligned)
        levels = ops.level_assign(rois[:, 1:].contiguous(), self.min_level, self.max_level,
                                  self.canonical_box_size, self.canonical_level)
        assert self.output_size[0] == self.output_size[1]
        return ops.roi_align_levels(x, self.scales, rois, levels, self.output_size[0], self.sampling_ratio,
                                    self.aligned)
