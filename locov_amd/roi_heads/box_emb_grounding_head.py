"""Multi-token class grounding predictor.

Mirrors ovr/modeling/roi_heads/box_emb_grounding_head.py: `GroundingModule` (:60-256) scores a region
embedding against class names of SEVERAL tokens each -- token similarities, masked softmax (or hardmax)
attention over a class's tokens, attention-weighted distance -- and
`EmbeddingGroundingFastRCNNOutputLayers` (:259-434) is the box predictor built on it.  In the reference
the predictor is unreachable with the shipped configs (`cfg.MODEL.ROI_HEADS.MAX_TOKENS`, read at :355, is
never defined); `locov_amd.config.get_cfg()` defines the key, which makes the yaml name usable here.

Device arithmetic: the token-similarity Linear is the f32 MFMA NT GEMM (`ops.linear`), everything
after it is one kernel (`locov_token_attention_fwd`), instead of a per-class Python loop over padded tensors.
Inference only (the reference builds it with frozen class tokens; the LSM configs never train through it).
"""
from __future__ import annotations

from typing import Dict, Union

import torch
from torch import nn

from .. import ops
from ..registry import configurable
from ..structures import ShapeSpec
from .box_emb_head import Box2BoxTransform, FastRCNNOutputLayers, _rownorm, hip_linear

__all__ = ["GroundingModule", "EmbeddingGroundingFastRCNNOutputLayers"]


class GroundingModule(nn.Module):
    """box_emb_grounding_head.py:60 -- same constructor, attributes (`class_emb`, `num_tok`, `mask_emb`,
    `token_score`) and `set_class_embeddings(embs: {class_idx: [n_tok, D]}, device)` contract."""

    def __init__(self, emb_dim, num_classes, max_tokens, local_metric: str = "dot",
                 global_metric: str = "aligned_local", alignment: str = "softmax", temperature: float = 1.0,
                 normalize_emb: bool = False, background_class: bool = True, return_similarity: bool = False):
        super().__init__()
        if local_metric not in ("dot", "cosine"):
            raise NotImplementedError(f"local_metric {local_metric!r}")
        if global_metric != "aligned_local":
            raise NotImplementedError(f"global_metric {global_metric!r}")
        if alignment not in ("softmax", "hardmax"):
            raise NotImplementedError(f"alignment {alignment!r}")
        if local_metric == "cosine":
            assert normalize_emb                                       # :96-97
        self.emb_dim, self.num_classes, self.max_tokens = emb_dim, num_classes, max_tokens
        self.local_metric, self.global_metric, self.alignment = local_metric, global_metric, alignment
        self.temperature = temperature
        self.normalize_emb = normalize_emb
        self.background_class = background_class
        self.return_similarity = return_similarity
        self.class_emb = torch.zeros(num_classes, emb_dim)
        self.num_tok = torch.ones(num_classes).int()
        self.mask_emb = torch.zeros(num_classes, 1)
        self.token_score = None
        self._tok_off = self._tok_cnt = self._split = None

    # ------------------------------------------------------------------ bank install (:227-256)
    def set_class_embeddings(self, embs: Dict[int, Union[torch.Tensor, list]], device):
        self.num_classes = len(embs)
        k1 = self.num_classes + (1 if self.background_class else 0)
        counts = [0] * k1
        pieces = []
        for cls_idx, cls_emb in embs.items():
            cls_emb = (cls_emb.clone().detach() if torch.is_tensor(cls_emb) else torch.tensor(cls_emb)).to(device, torch.float32)
            counts[cls_idx] = cls_emb.shape[0]
            pieces.append(cls_emb)
        if self.background_class:
            pieces.append(torch.zeros(1, self.emb_dim, device=device))     # the background token (:242-243)
        self.num_tok = torch.tensor(counts, dtype=torch.int32, device=device)
        tmax = max(max(counts), 1)
        self.mask_emb = (torch.arange(tmax, device=device)[None, :] < self.num_tok[:, None]).to(torch.float32)
        self.class_emb = torch.cat(pieces, 0)
        if self.normalize_emb:
            assert self.class_emb.shape[1] == self.emb_dim, "The embedding dimension has to match the one saved in the model"
            self.class_emb = _rownorm(self.class_emb, ops.NORM_L2)
        self.token_score = nn.Linear(self.emb_dim, self.class_emb.shape[0]).to(device)
        self.token_score.weight.data = self.class_emb
        self.token_score.bias.data = torch.zeros_like(self.token_score.bias.data)
        self.token_score.weight.requires_grad = False
        self.token_score.bias.requires_grad = False
        # Column layout of token_score's output: classes in dict order, a token-less class owns one column
        # (the reference's split gives it one: `split_sizes[split_sizes == 0] = 1`, :129-131).
        order = list(embs.keys()) + ([k1 - 1] if self.background_class else [])
        width = [max(counts[k], 1) if (k in embs or k == k1 - 1) else 0 for k in range(k1)]
        off, pos = [0] * k1, 0
        for k in order:
            off[k] = pos
            pos += counts[k] if k in embs else 1
        assert pos == self.class_emb.shape[0]
        self._split = [max(counts[k], 1) for k in order]
        self._tok_off = torch.tensor(off, dtype=torch.int32, device=device)
        self._tok_cnt = self.num_tok.clone()
        self._padded = any(w < tmax for w in width)

    # ------------------------------------------------------------------ forward (:199-225)
    @torch.no_grad()
    def forward(self, image_emb: torch.Tensor):
        """image_emb [R, D] -> (scores [R, K+1] = -global distance, token attention [R, K+1, Tmax])."""
        if self.token_score is None:
            raise RuntimeError("set_class_embeddings() must be called before forward()")
        sim = ops.linear(image_emb.detach(), self.token_score.weight, self.token_score.bias)      # :103 / :106
        cosine = self.local_metric == "cosine"
        s = torch.nan_to_num(sim, nan=0.0, posinf=float("inf"), neginf=float("-inf")) if cosine else sim
        # minimum of the reference's zero-padded [R, K+1, Tmax] similarity tensor (:167-169)
        gmin = (s / self.temperature).min() if s.numel() else s.new_zeros(())
        if self._padded:
            gmin = torch.minimum(gmin, gmin.new_zeros(()))
        tmax = self.mask_emb.shape[1]
        scores, att = ops.token_attention(sim, self._tok_off, self._tok_cnt, tmax, self.temperature, gmin,
                                          cosine=cosine, hardmax=self.alignment == "hardmax")
        if self.return_similarity:
            loc_sim = s / self.temperature
            loc_dis = ((1 - s) if cosine else -s) / self.temperature
            return scores, att, (torch.split(loc_sim, self._split, dim=1), torch.split(loc_dis, self._split, dim=1))
        return scores, att


class EmbeddingGroundingFastRCNNOutputLayers(FastRCNNOutputLayers):
    """box_emb_grounding_head.py:259 -- `emb_pred` + `bbox_pred` + a GroundingModule as `cls_score`."""

    @configurable
    def __init__(self, input_shape, *, box2box_transform, num_classes: int, test_score_thresh: float = 0.0,
                 test_nms_thresh: float = 0.5, test_topk_per_image: int = 100, cls_agnostic_bbox_reg: bool = False,
                 smooth_l1_beta: float = 0.0, box_reg_loss_type: str = "smooth_l1",
                 loss_weight: Union[float, Dict[str, float]] = 1.0, emb_dim: int = 768, embedding_based: bool = True,
                 freeze_emb_pred: bool = True, normalize_emb: bool = False, detach_cls_predictor: bool = False,
                 grounding_module: GroundingModule = None):
        FastRCNNOutputLayers.__init__(
            self, input_shape, box2box_transform=box2box_transform, num_classes=num_classes,
            test_score_thresh=test_score_thresh, test_nms_thresh=test_nms_thresh,
            test_topk_per_image=test_topk_per_image, cls_agnostic_bbox_reg=cls_agnostic_bbox_reg,
            smooth_l1_beta=smooth_l1_beta, box_reg_loss_type=box_reg_loss_type, loss_weight=loss_weight)
        if isinstance(input_shape, int):
            input_shape = ShapeSpec(channels=input_shape)
        num_inputs = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        self.embedding_based = embedding_based
        if self.embedding_based:
            self.normalize_emb = normalize_emb
            self.emb_dim = emb_dim
            self.emb_pred = nn.Linear(num_inputs, self.emb_dim)
            nn.init.normal_(self.emb_pred.weight, mean=0, std=0.01)       # :337
            nn.init.constant_(self.emb_pred.bias, 0)
            self.cls_score = grounding_module
            assert cls_agnostic_bbox_reg
            self.num_classes = None                                       # until set_class_embeddings (:341-342)
        self.detach_cls_predictor = detach_cls_predictor
        if self.detach_cls_predictor:
            self.loss_weight.update({"loss_cls": 0.0})                    # :346-348

    @classmethod
    def from_config(cls, cfg, input_shape):
        ret = FastRCNNOutputLayers.from_config.__func__(cls, cfg, input_shape)
        g = cfg.MODEL.MMSS_HEAD.GROUNDING
        box_head = cfg.MODEL.ROI_BOX_HEAD
        ret.update({
            "emb_dim": box_head.EMB_DIM,
            "embedding_based": box_head.EMBEDDING_BASED,
            "freeze_emb_pred": box_head.FREEZE_EMB_PRED,
            "normalize_emb": box_head.NORMALIZE_EMB_PRED,
            "detach_cls_predictor": cfg.MODEL.ROI_HEADS.DETACH_CLASS_PREDICTOR,
            "grounding_module": GroundingModule(
                box_head.EMB_DIM, cfg.MODEL.ROI_HEADS.NUM_CLASSES, cfg.MODEL.ROI_HEADS.MAX_TOKENS,
                local_metric=g.LOCAL_METRIC, global_metric=g.GLOBAL_METRIC, alignment=g.ALIGNMENT,
                temperature=g.ALIGNMENT_TEMPERATURE, normalize_emb=box_head.NORMALIZE_EMB_PRED),
        })
        return ret

    def device(self):
        return self.emb_pred.weight.device

    def forward(self, x):
        """(scores [R,K+1], proposal_deltas [R,4]) (:395-417)."""
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        proposal_deltas = hip_linear(x, self.bbox_pred)
        if self.detach_cls_predictor:
            with torch.no_grad():
                scores = self.forward_cls_prediction(x.detach())
        else:
            scores = self.forward_cls_prediction(x)
        return scores, proposal_deltas

    def forward_cls_prediction(self, x):                                  # :419-427
        if not self.embedding_based:
            return self.cls_score(x)
        x = hip_linear(x, self.emb_pred)
        if self.normalize_emb:
            x = _rownorm(x, ops.NORM_L2)
        scores, _ = self.cls_score(x)
        return scores

    def set_class_embeddings(self, embs):                                 # :429-431
        self.cls_score.set_class_embeddings(embs, self.device())
        self.num_classes = self.cls_score.num_classes
