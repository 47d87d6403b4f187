"""Stand-alone equivalents of the Detectron2 structures the LSM ROI head exchanges with its
callers (SURVEY.md 8b "Call signature"): Boxes, Instances, ImageList, ShapeSpec.

[D2-upstream] detectron2.structures / detectron2.layers.ShapeSpec: same attribute and method
names for everything the reference touches (ovr/modeling/roi_heads/roi_emb_heads.py:25-118,
:258-282, :311-360; distill_prop_mmss_gcnn.py:348-417).  When Detectron2 itself is
importable its own classes are accepted as well (duck typing on .tensor / .get_fields()).
"""
from __future__ import annotations

import itertools
from collections import namedtuple
from typing import Any, Dict, List, Sequence, Tuple, Union

import torch


class ShapeSpec(namedtuple("_ShapeSpec", ["channels", "height", "width", "stride"])):
    def __new__(cls, channels=None, height=None, width=None, stride=None):
        return super().__new__(cls, channels, height, width, stride)


class Boxes:
    """[N,4] XYXY absolute fp32 boxes."""

    def __init__(self, tensor: torch.Tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        tensor = tensor.to(torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4))
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self) -> "Boxes":
        return Boxes(self.tensor.clone())

    def to(self, *args, **kwargs) -> "Boxes":
        return Boxes(self.tensor.to(*args, **kwargs))

    def area(self) -> torch.Tensor:
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size: Tuple[int, int]) -> None:
        h, w = box_size
        x1 = self.tensor[:, 0].clamp(min=0, max=w)
        y1 = self.tensor[:, 1].clamp(min=0, max=h)
        x2 = self.tensor[:, 2].clamp(min=0, max=w)
        y2 = self.tensor[:, 3].clamp(min=0, max=h)
        self.tensor = torch.stack((x1, y1, x2, y2), dim=-1)

    def nonempty(self, threshold: float = 0.0) -> torch.Tensor:
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def get_centers(self) -> torch.Tensor:
        return (self.tensor[:, :2] + self.tensor[:, 2:]) / 2

    def scale(self, scale_x: float, scale_y: float) -> None:
        """[D2-upstream] Boxes.scale (in place): what detector_postprocess applies to pred_boxes
        (ovr/modeling/meta_arch/ovr_rcnn.py:111, distill_prop_mmss_gcnn.py:556)."""
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    def inside_box(self, box_size: Tuple[int, int], boundary_threshold: int = 0) -> torch.Tensor:
        height, width = box_size
        b = self.tensor
        return ((b[..., 0] >= -boundary_threshold) & (b[..., 1] >= -boundary_threshold)
                & (b[..., 2] < width + boundary_threshold) & (b[..., 3] < height + boundary_threshold))

    def __getitem__(self, item) -> "Boxes":
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2, f"Indexing on Boxes with {item} failed to return a matrix!"
        return Boxes(b)

    def __len__(self) -> int:
        return self.tensor.shape[0]

    def __repr__(self) -> str:
        return "Boxes(" + str(self.tensor) + ")"

    @property
    def device(self):
        return self.tensor.device

    @classmethod
    def cat(cls, boxes_list: Sequence["Boxes"]) -> "Boxes":
        if len(boxes_list) == 0:
            return cls(torch.empty(0, 4))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    def __iter__(self):
        yield from self.tensor


def pairwise_iou(boxes1: Boxes, boxes2: Boxes) -> torch.Tensor:
    """[D2-upstream] pairwise_iou: [N,M] IoU (0 where the intersection is empty)."""
    a, b = boxes1.tensor, boxes2.tensor
    area1, area2 = boxes1.area(), boxes2.area()
    wh = torch.min(a[:, None, 2:], b[:, 2:]) - torch.max(a[:, None, :2], b[:, :2])
    wh.clamp_(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (area1[:, None] + area2 - inter),
                       torch.zeros(1, dtype=inter.dtype, device=inter.device))


class Instances:
    """Per-image container of same-length fields (Detectron2 Instances semantics)."""

    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        self._image_size = image_size
        self._fields: Dict[str, Any] = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self) -> Tuple[int, int]:
        return self._image_size

    def __setattr__(self, name: str, val: Any) -> None:
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name: str) -> Any:
        if name == "_fields" or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def set(self, name: str, value: Any) -> None:
        data_len = len(value)
        if len(self._fields):
            assert len(self) == data_len, f"Adding a field of length {data_len} to a Instances of length {len(self)}"
        self._fields[name] = value

    def has(self, name: str) -> bool:
        return name in self._fields

    def remove(self, name: str) -> None:
        del self._fields[name]

    def get(self, name: str) -> Any:
        return self._fields[name]

    def get_fields(self) -> Dict[str, Any]:
        return self._fields

    def to(self, *args, **kwargs) -> "Instances":
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            if hasattr(v, "to"):
                v = v.to(*args, **kwargs)
            ret.set(k, v)
        return ret

    def __getitem__(self, item) -> "Instances":
        if isinstance(item, int):
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self) -> int:
        for v in self._fields.values():
            return len(v)
        raise NotImplementedError("Empty Instances does not support __len__!")

    @staticmethod
    def cat(instance_lists: List["Instances"]) -> "Instances":
        assert len(instance_lists) > 0
        if len(instance_lists) == 1:
            return instance_lists[0]
        image_size = instance_lists[0].image_size
        ret = Instances(image_size)
        for k in instance_lists[0]._fields.keys():
            values = [i.get(k) for i in instance_lists]
            v0 = values[0]
            if isinstance(v0, torch.Tensor):
                values = torch.cat(values, dim=0)
            elif isinstance(v0, list):
                values = list(itertools.chain(*values))
            elif hasattr(type(v0), "cat"):
                values = type(v0).cat(values)
            else:
                raise ValueError(f"Unsupported type {type(v0)} for concatenation")
            ret.set(k, values)
        return ret

    def __repr__(self) -> str:
        s = self.__class__.__name__ + "("
        s += f"num_instances={len(self) if self._fields else 0}, "
        s += f"image_height={self._image_size[0]}, image_width={self._image_size[1]}, "
        s += "fields=[{}])".format(", ".join(f"{k}: {v}" for k, v in self._fields.items()))
        return s


class ImageList:
    """Batched image tensor + per-image (h, w); the ROI heads only `del` it
    (roi_emb_heads.py:251,315) but callers construct one."""

    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self) -> int:
        return len(self.image_sizes)

    @property
    def device(self):
        return self.tensor.device


def cat_rows(tensors: Sequence[torch.Tensor]) -> torch.Tensor:
    """torch.cat(tensors, dim=0) -- as a VIEW, without a launch, when the pieces are contiguous, consecutive row ranges of one
    storage (the per-image views of a batch-wide gather: what the sampled Instances of a training step hold)."""
    tensors = list(tensors)
    if len(tensors) == 1:
        return tensors[0]
    if tensors and tensors[0].dim() >= 1:
        t0 = tensors[0]
        store, ptr, total = t0.untyped_storage().data_ptr(), t0.data_ptr(), 0
        for t in tensors:
            if (t.requires_grad or t.dtype != t0.dtype or t.shape[1:] != t0.shape[1:] or not t.is_contiguous() or t.data_ptr() != ptr
                    or t.untyped_storage().data_ptr() != store):
                break
            ptr += t.numel() * t.element_size()
            total += t.shape[0]
        else:
            return t0.as_strided((total,) + tuple(t0.shape[1:]), t0.stride())
    return torch.cat(tensors, dim=0)


def boxes_class_of(proposals: Sequence[Any]):
    """(Instances class, Boxes class) of the caller's objects: results are built with the SAME classes the caller handed in,
    so that under Detectron2 the meta-architecture gets detectron2.structures back (detector_postprocess calls
    pred_boxes.scale / .clip / .nonempty on them) and stand-alone callers get this module's."""
    for p in proposals:
        inst_cls = type(p)
        for name in ("proposal_boxes", "gt_boxes", "pred_boxes"):
            if p.has(name):
                return inst_cls, type(p.get(name))
        return inst_cls, Boxes
    return Instances, Boxes


def boxes_tensor(b: Union[Boxes, torch.Tensor, Any]) -> torch.Tensor:
    """Accepts our Boxes, a Detectron2 Boxes (has .tensor) or a raw [N,4] tensor."""
    return b.tensor if hasattr(b, "tensor") else b
