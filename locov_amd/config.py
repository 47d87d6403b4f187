"""The configuration keys the LSM ROI-head path reads, with the reference's defaults.

The reference uses a yacs CfgNode (detectron2.config.get_cfg + ovr/config/config.py:4-174
add_ovr_config).  yacs / Detectron2 are not installed here, so this is a minimal attribute
tree holding exactly the keys of SURVEY.md section 5.6 -- same names, same defaults
([D2-upstream] defaults marked) -- and it can read the reference's own yaml files
(configs/coco_lsm.yaml, configs/coco_stt.yaml): unknown keys are kept, not rejected, so the
rest of those files passes through untouched.  A real Detectron2 CfgNode works just as well
with every from_config in this package (only attribute access is used).
"""
from __future__ import annotations

import copy
from ast import literal_eval
from typing import Any, Dict


class CfgNode(dict):
    def __init__(self, init: Dict[str, Any] = None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def clone(self) -> "CfgNode":
        return copy.deepcopy(self)

    def merge_from_dict(self, other: Dict[str, Any]) -> None:
        for k, v in other.items():
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], CfgNode):
                    self[k] = CfgNode()
                self[k].merge_from_dict(v)
            else:
                if isinstance(v, str):
                    try:    # yaml keeps python tuples such as ("backbone.body.",) as strings
                        lit = literal_eval(v)
                        if isinstance(lit, (tuple, list)):
                            v = lit
                    except (ValueError, SyntaxError):
                        pass
                self[k] = v

    def merge_from_file(self, path: str) -> None:
        import yaml
        with open(path, "r") as f:
            self.merge_from_dict(yaml.safe_load(f) or {})

    def merge_from_list(self, opts) -> None:
        """["MODEL.ROI_HEADS.NUM_CLASSES", 80, ...] as in train_ovnet.py:49-56."""
        assert len(opts) % 2 == 0
        for full_key, v in zip(opts[0::2], opts[1::2]):
            node = self
            *parents, leaf = full_key.split(".")
            for p in parents:
                node = node.setdefault(p, CfgNode())
            if isinstance(v, str):
                try:
                    v = literal_eval(v)
                except (ValueError, SyntaxError):
                    pass
            node[leaf] = v


def get_cfg() -> CfgNode:
    """Defaults for the keys the hot path reads."""
    return CfgNode({
        "MODEL": {
            "DEVICE": "cuda",
            "MASK_ON": False,
            "KEYPOINT_ON": False,
            "LOAD_EMB_PRED_FROM_MMSS_HEAD": False,                # config.py:13
            "ROI_HEADS": {
                "NAME": "EmbeddingRes5ROIHeads",
                "IN_FEATURES": ["res4"],                           # [D2-upstream]
                "NUM_CLASSES": 80,
                "BATCH_SIZE_PER_IMAGE": 512,                       # [D2-upstream]; coco_lsm.yaml:32 -> 200
                "POSITIVE_FRACTION": 0.25,                         # [D2-upstream]; coco_lsm.yaml:30 -> 1.0
                "IOU_THRESHOLDS": [0.5],                           # [D2-upstream]
                "IOU_LABELS": [0, 1],                              # [D2-upstream]
                "PROPOSAL_APPEND_GT": True,                        # [D2-upstream]
                "SCORE_THRESH_TEST": 0.05,                         # [D2-upstream]
                "NMS_THRESH_TEST": 0.5,                            # [D2-upstream]
                "DETACH_CLASS_PREDICTOR": False,                   # config.py:136
                # read by EmbeddingGroundingFastRCNNOutputLayers.from_config (box_emb_grounding_head.py:355) but
                # never defined in ovr/config/config.py; defined here so that the predictor can be built
                "MAX_TOKENS": 8,
            },
            "MMSS_HEAD": {
                "GROUNDING": {                                     # config.py:51-55 (keys the box predictor reads)
                    "LOCAL_METRIC": "dot", "GLOBAL_METRIC": "aligned_local", "ALIGNMENT": "softmax",
                    "ALIGNMENT_TEMPERATURE": 10.0,
                },
            },
            "ROI_BOX_HEAD": {
                "NAME": "EmbeddingFastRCNNOutputLayers",
                "POOLER_RESOLUTION": 14,                           # [D2-upstream]
                "POOLER_TYPE": "ROIAlignV2",                       # [D2-upstream]
                "POOLER_SAMPLING_RATIO": 0,                        # [D2-upstream]
                "CLS_AGNOSTIC_BBOX_REG": False,                    # coco_lsm.yaml:36 -> True
                "BBOX_REG_WEIGHTS": (10.0, 10.0, 5.0, 5.0),        # [D2-upstream]
                "BBOX_REG_LOSS_TYPE": "smooth_l1",                 # [D2-upstream]
                "BBOX_REG_LOSS_WEIGHT": 1.0,                       # [D2-upstream]
                "SMOOTH_L1_BETA": 0.0,                             # [D2-upstream]
                "EMBEDDING_BASED": False,                          # config.py:124
                "EMB_DIM": 768,                                    # config.py:126
                "FREEZE_EMB_PRED": False,                          # config.py:129
                "NORMALIZE_EMB_PRED": False,                       # config.py:131
                "STANDARDIZE_EMB_PRED": False,                     # config.py:133
                # extension (not in the reference): dtype of the similarity GEMM operands
                "SIM_GEMM_DTYPE": "fp32",
                # extension: "hip" = hand-written channels-last MFMA GEMM Res5, "miopen" = torch conv2d
                "RES5_BACKEND": "hip",
                "RES5_CONV3X3": "winograd",
                # extension: arithmetic of the Res5 GEMMs on the HIP backend.  "f16x2": fp32 in / fp32 out, products formed
                # from split (hi, lo) f16 operand pairs on the f16 matrix pipe, fp32 accumulate -- error vs fp64 no larger
                # than the f32 MFMA's, ~2x its speed (activations must stay below 4094 in magnitude); "fp32": the f32 MFMA;
                # "bf16": bf16 operands (reduced precision, opt-in, not a parity configuration)
                "RES5_DTYPE": "f16x2",
                # extension: with "f16x2" every split launch ORs a range-guard word when an activation left fp16's range; the
                # heads read it once per call and repeat the call on the f32 MFMA (with a warning) if it is set
                "RES5_OVERFLOW_CHECK": True,
                # training forwards: "sync" = the guard word is looked at at the end of the ROI heads' forward (an event wait behind
                # the Res5 calls that leaves the predictor's launches queued) and an out-of-range forward is repeated on the f32
                # MFMA; "deferred" = never read inside the step, acted on on the device (a skipped step), read with the next labelling
                "RES5_TRAIN_GUARD": "sync",
            },
            "RESNETS": {
                "NUM_GROUPS": 1, "WIDTH_PER_GROUP": 64, "RES2_OUT_CHANNELS": 256,
                "STRIDE_IN_1X1": True, "NORM": "FrozenBN",
                "DEFORM_ON_PER_STAGE": [False, False, False, False],
            },
        },
        "TEST": {"DETECTIONS_PER_IMAGE": 100},                     # [D2-upstream]
    })
