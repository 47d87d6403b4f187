"""Text bank of noun embeddings: the data format on the input side of the similarity GEMM
(SURVEY.md 8f-2).

On disk the reference keeps `{class_name: [D floats]}` JSON written by
tools/coco_bert_embeddings.py:16-38 (mean of BERT input word embeddings over the non-special
tokens of the class name).  Dataset registration turns it into `class_emb_mtx`, a float32
[K+1, D] matrix in `thing_classes` order whose LAST row is the all-zero background class
(ovr/data/datasets/coco_instances.py:228-254, lvis_instances.py:260-278), and
`OVRTrainer.load_embeddings` installs it with `box_predictor.set_class_embeddings`
(ovr/engine/trainer.py:365-396), re-installing a different bank per test dataset during
evaluation (:187-191, :254-257; 48 / 17 / 65 classes for COCO base / novel / all).

This module provides exactly those steps plus the MI355X-side packing: a resident fp32 copy and
(optionally) a bf16 copy for the bf16-MFMA similarity GEMM, cached per dataset so that swapping
banks during evaluation is a pointer change, not a re-upload.
"""
from __future__ import annotations

import json
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

__all__ = ["load_noun_embeddings", "build_class_emb_mtx", "build_lvis_class_emb_mtx", "pool_token_embeddings",
           "TextBankCache"]


def load_noun_embeddings(path: str) -> Dict[str, np.ndarray]:
    """`{class_name: [D floats]}` JSON (a multi-token class may hold `[T][D]`) -> dict of float32 arrays."""
    with open(path, "r") as f:
        raw = json.load(f)
    return {k: np.asarray(v, dtype=np.float32) for k, v in raw.items()}


def build_class_emb_mtx(noun_embeddings: Dict[str, Sequence[float]], thing_classes: Sequence[str],
                        return_class_embeddings: bool = False):
    """COCO layout, coco_instances.py:237-254: [len(thing_classes)+1, D] float32, classes in `thing_classes` order,
    last row = zeros (background).  D is the length of the FIRST entry of the file (:238).  A class whose entry is
    not a vector (a multi-token `[T, D]` embedding for the grounding predictor, box_emb_grounding_head.py) keeps a ZERO
    row (:247-250) and, with `return_class_embeddings`, the per-class dict `{idx: array}` the reference then attaches
    to the metadata (:252-253) is returned as well (None when every class is a plain vector).  A class missing from
    the file is a KeyError, as in the reference."""
    first = next(iter(noun_embeddings.values()))
    emb_dim = len(first)
    mtx = np.zeros((len(thing_classes) + 1, emb_dim), dtype=np.float32)
    class_embeddings, multi = {}, False
    for idx, noun in enumerate(thing_classes):
        vec = np.asarray(noun_embeddings[noun], dtype=np.float32)
        class_embeddings[idx] = vec
        if vec.ndim == 1:
            if vec.shape[0] != emb_dim:
                raise ValueError(f"embedding of {noun!r} has shape {vec.shape}, expected ({emb_dim},)")
            mtx[idx, :] = vec
        else:
            multi = True
    if return_class_embeddings:
        return mtx, (class_embeddings if multi else None)
    return mtx


def build_lvis_class_emb_mtx(noun_embeddings: Optional[Dict[str, Sequence[float]]], thing_classes: Sequence[str]):
    """LVIS layout, lvis_instances.py:262-278: the bank is optional (no `obj_file` -> no class_emb_mtx: returns None);
    every class must be a plain vector (the reference assigns the entry straight into its row, :277, so a multi-token
    entry is a broadcast error there and a ValueError here)."""
    if noun_embeddings is None:
        return None
    emb_dim = len(next(iter(noun_embeddings.values())))
    mtx = np.zeros((len(thing_classes) + 1, emb_dim), dtype=np.float32)
    for idx, noun in enumerate(thing_classes):
        vec = np.asarray(noun_embeddings[noun], dtype=np.float32)
        if vec.shape != (emb_dim,):
            raise ValueError(f"could not broadcast the embedding of {noun!r}, shape {vec.shape}, into a row of ({emb_dim},)")
        mtx[idx, :] = vec
    return mtx


def pool_token_embeddings(input_embeddings: np.ndarray, special_tokens_mask: np.ndarray) -> np.ndarray:
    """tools/coco_bert_embeddings.py:26-30: mean of the token embeddings over the non-special tokens.
    input_embeddings [K, T, D], special_tokens_mask [K, T] (1 = special) -> [K, D]."""
    mask = (1 - np.asarray(special_tokens_mask)).astype(np.float32)
    emb = np.asarray(input_embeddings, dtype=np.float32)
    return (emb * mask[:, :, None]).sum(1) / mask.sum(1)[:, None]


class TextBankCache:
    """Device-resident banks keyed by dataset name; `install(name, predictor)` is what
    `load_embeddings` does for that dataset."""

    def __init__(self, device="cuda", with_bf16: bool = False):
        self.device = torch.device(device)
        self.with_bf16 = with_bf16
        self._fp32: Dict[str, torch.Tensor] = {}
        self._bf16: Dict[str, torch.Tensor] = {}

    def add(self, name: str, class_emb_mtx) -> torch.Tensor:
        mtx = torch.as_tensor(np.asarray(class_emb_mtx, dtype=np.float32))
        if mtx.dim() != 2 or not torch.all(mtx[-1] == 0):
            raise ValueError("class_emb_mtx must be [K+1, D] with an all-zero background row last")
        self._fp32[name] = mtx.to(self.device).contiguous()
        if self.with_bf16:
            from . import ops
            self._bf16[name] = ops.to_bf16(self._fp32[name])
        return self._fp32[name]

    def add_from_json(self, name: str, path: str, thing_classes: Sequence[str]) -> torch.Tensor:
        return self.add(name, build_class_emb_mtx(load_noun_embeddings(path), thing_classes))

    def names(self) -> List[str]:
        return list(self._fp32)

    def get(self, name: str, bf16: bool = False) -> Optional[torch.Tensor]:
        return (self._bf16 if bf16 else self._fp32)[name]

    def install(self, name: str, box_predictor, roi_heads=None) -> None:
        """trainer.py:383-396: set_class_embeddings + propagate num_classes to the ROI heads."""
        box_predictor.set_class_embeddings(self._fp32[name])
        if self.with_bf16 and not (box_predictor.normalize_emb or box_predictor.standardize_emb):
            box_predictor.install_packed_bank(self._bf16[name])   # pre-packed: no per-swap conversion
        if roi_heads is not None and hasattr(roi_heads, "num_classes"):
            roi_heads.num_classes = box_predictor.num_classes
