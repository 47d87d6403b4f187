"""GroundingHead of the LSM branch on the gfx950 kernels.

Mirrors ovr/modeling/mmss_heads/grounding_head.py:51-392 (class, constructor arguments, config
keys, `v2l_projection` parameter name, forward signature, returned dictionaries and their keys)
for the combination configs/coco_lsm.yaml selects: LOCAL_METRIC "dot", GLOBAL_METRIC
"aligned_local", ALIGNMENT "softmax", LOSS "cross_entropy".  The reference raises
NotImplementedError for metrics it does not define; the remaining variants it does define
(hardmax / random alignments, reconstruction_mse, triplet / matching losses) are not on the LSM
path and raise NotImplementedError here.

What changes is HOW: instead of materialising B^2 copies of the captions, regions and masks
(:119-144) and running ~30 elementwise launches on [B^2, T, NR] tensors with a host sync per logged
tensor (`LoggedModule.log`), the region embeddings and the caption tokens meet in ONE [B*T, B*NR]
GEMM and one fused kernel reduces each T x NR block to its two cost entries (csrc/grounding.hip).
The B x B cross-entropy tail stays on torch ops (a few dozen scalars).
"""
from __future__ import annotations

from typing import Dict

import torch
from torch import nn

from . import ops

__all__ = ["GroundingHead", "build_grounding_head"]


def _get(cfg, name, default):
    return getattr(cfg, name) if hasattr(cfg, name) else default


class GroundingHead(nn.Module):
    def __init__(self, config, v_dim, l_dim, *args, **kwargs):
        super().__init__()
        g = config.MODEL.MMSS_HEAD.GROUNDING
        self.config = g
        self.v_dim, self.l_dim = v_dim, l_dim
        self.v2l_projection = nn.Linear(self.v_dim, self.l_dim)
        self.local_metric = g.LOCAL_METRIC
        self.global_metric = g.GLOBAL_METRIC
        self.alignment = g.ALIGNMENT
        self.temperature = g.ALIGNMENT_TEMPERATURE
        self.loss_type = g.LOSS
        self.negative_mining = _get(g, "NEGATIVE_MINING", "random")
        self.margin = _get(g, "TRIPLET_MARGIN", 1.0)
        self.align_words = g.ALIGN_WORDS_TO_REGIONS
        self.align_regions = g.ALIGN_REGIONS_TO_WORDS
        assert self.align_words or self.align_regions
        self.return_dist = config.MODEL.MMSS_HEAD.DISTILLATION_LOSS
        self.grounding_text_input = _get(g, "TEXT_INPUT", "input_embeddings")
        self.log_info: Dict[str, object] = {}          # LoggedModule.log_info (filled lazily, no host syncs)
        if (self.local_metric, self.global_metric, self.alignment, self.loss_type) != \
                ("dot", "aligned_local", "softmax", "cross_entropy"):
            raise NotImplementedError(
                "the MI355X GroundingHead implements the LSM configuration only: LOCAL_METRIC=dot, "
                "GLOBAL_METRIC=aligned_local, ALIGNMENT=softmax, LOSS=cross_entropy")

    def forward(self, input_image, input_caption):
        caption_emb = input_caption[self.grounding_text_input]                       # [B, T, L]
        caption_mask = (input_caption["attention_mask"] * (1 - input_caption["special_tokens_mask"])
                        ).to(torch.float32)                                          # :94-96
        region_features = input_image["region_features"]                             # [B, NR, V]
        region_mask = input_image["region_mask"].to(torch.float32)
        B, NR, V = region_features.shape
        T = caption_mask.shape[1]
        num_words = caption_mask.sum(dim=1)
        num_regions = region_mask.sum(dim=1)

        # :111 image_emb = v2l_projection(region_features); weights read at call time (tied to emb_pred)
        image_emb = ops.linear_autograd(region_features.reshape(B * NR, V).contiguous().float(),
                                        self.v2l_projection.weight, self.v2l_projection.bias)   # [B*NR, L]
        cap = caption_emb.reshape(B * T, -1).contiguous().float()
        # :147 all B^2 caption x image token-region similarities as one NT GEMM
        S = ops.linear_autograd(cap, image_emb, None)                                 # [B*T, B*NR]
        # :150-228 temperature, masked softmax both ways, aligned-local distances -> [caption, image] costs
        cost_w2r, cost_r2w = ops.grounding_costs(S, caption_mask, region_mask, self.temperature)
        # :232-243 pairs with neither words nor regions get (max + 100)
        ok = (num_words[:, None] > 0) | (num_regions[None, :] > 0)
        losses, other_info = {}, {}
        eye = torch.arange(B, device=S.device)
        pw = {}
        for on, tag, cost in ((self.align_words, "Words", cost_w2r), (self.align_regions, "Regions", cost_r2w)):
            if not on:
                continue
            cost = torch.where(ok, cost, cost.max().detach() + 100.0)
            pw[tag] = cost
            logits_cap = torch.log_softmax(-cost, dim=0)                              # :264-277
            logits_img = torch.log_softmax(-cost, dim=1)
            losses[f"CE_loss (Align {tag}, Choose Caption)"] = torch.diag(-logits_cap).mean()
            losses[f"CE_loss (Align {tag}, Choose Image)"] = torch.diag(-logits_img).mean()
            other_info[f"Batch Accuracy (Align {tag}, Choose Caption)"] = (cost.argmin(dim=0) == eye).float().mean()
            other_info[f"Batch Accuracy (Align {tag}, Choose Image)"] = (cost.argmin(dim=1) == eye).float().mean()
        self.log_info = {**losses, **other_info}
        if self.return_dist:
            return other_info, losses, {"w2r": pw.get("Words"), "r2w": pw.get("Regions")}
        return other_info, losses


def build_grounding_head(name, cfg, v_dim, l_dim, *args, **kwargs):
    if name != "GroundingHead":
        raise KeyError(f"No object named '{name}' found in 'MMSS_HEADS' registry!")
    return GroundingHead(cfg, v_dim, l_dim)
