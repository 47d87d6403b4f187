"""[D2-upstream] detectron2.modeling.roi_heads: the registry train_ovnet.py's build_roi_heads resolves
MODEL.ROI_HEADS.NAME in (test stub)."""
from detectron2.utils.registry import Registry

ROI_HEADS_REGISTRY = Registry("ROI_HEADS")


def build_roi_heads(cfg, input_shape):
    name = cfg.MODEL.ROI_HEADS.NAME
    return ROI_HEADS_REGISTRY.get(name)(cfg, input_shape)
