"""locov_amd -- MI355X (gfx950) implementation of LocOV's Localized-Semantic-Matching ROI head.

Only what the hot path needs: hand-written HIP kernels behind a C ABI
(csrc/, include/locov_hip.h), their ctypes binding (_lib, ops), and the host-side mirror of the
reference's Detectron2 plugin surface (poolers, res5, roi_heads).
"""
__version__ = "0.1.0"

from . import config, ops, poolers, res5, structures  # noqa: F401
from .roi_heads import (EmbeddingFastRCNNOutputLayers, EmbeddingProposalsRes5ROIHeads,  # noqa: F401
                        EmbeddingRes5ROIHeads, build_box_predictor, build_roi_heads)
