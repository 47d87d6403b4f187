"""ROI heads / box predictor of the LSM path (same names as ovr/modeling/roi_heads).

Importing this package registers the heads in this package's ROI_HEADS_REGISTRY and, when
Detectron2 is importable, ALSO in detectron2's ROI_HEADS_REGISTRY under the reference's
names -- that is the drop-in: `MODEL.ROI_HEADS.NAME: "EmbeddingProposalsRes5ROIHeads"` in
configs/coco_lsm.yaml then resolves to the MI355X implementation.
"""
from .box_emb_head import (Box2BoxTransform, EmbeddingFastRCNNOutputLayers, FastRCNNOutputLayers,
                           build_box_predictor, fast_rcnn_inference)
from .box_emb_grounding_head import EmbeddingGroundingFastRCNNOutputLayers, GroundingModule
from .roi_emb_heads import (ROI_HEADS_REGISTRY, EmbeddingProposalsRes5ROIHeads, EmbeddingRes5ROIHeads,
                            SampleAllROIHeads, build_roi_heads)


def register_with_detectron2(override: bool = True) -> bool:
    """Put the MI355X heads into Detectron2's registry (replacing the reference's entries of the
    same name when `override`).  Returns False when Detectron2 is not installed."""
    try:
        from detectron2.modeling.roi_heads import ROI_HEADS_REGISTRY as D2_REGISTRY
    except Exception:
        return False
    for cls in (EmbeddingRes5ROIHeads, EmbeddingProposalsRes5ROIHeads):
        if cls.__name__ in D2_REGISTRY._obj_map:
            if not override:
                continue
            del D2_REGISTRY._obj_map[cls.__name__]
        D2_REGISTRY.register(cls)
    return True


__all__ = ["Box2BoxTransform", "EmbeddingGroundingFastRCNNOutputLayers", "GroundingModule", "EmbeddingFastRCNNOutputLayers", "FastRCNNOutputLayers", "build_box_predictor",
           "fast_rcnn_inference", "ROI_HEADS_REGISTRY", "EmbeddingProposalsRes5ROIHeads", "EmbeddingRes5ROIHeads",
           "SampleAllROIHeads", "build_roi_heads", "register_with_detectron2"]
