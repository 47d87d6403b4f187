import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import locov_amd as pkg
from oracle import lsm_oracle as oracle
import test_gpu_roi_heads as T
for dtype in ["fp32", "f16x2"]:
    cfg = pkg.config.get_cfg()
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    cfg.MODEL.ROI_HEADS.NAME = "EmbeddingProposalsRes5ROIHeads"
    cfg.MODEL.ROI_BOX_HEAD.RES5_DTYPE = dtype
    heads, params, h = T._make_heads(pkg, oracle, cfg, 80, 1992)
    rng = np.random.default_rng(1992)
    feat = rng.standard_normal((2, 1024, 50, 84)).astype(np.float32)
    props, boxes = T._proposals(pkg, oracle, rng, 2, 40)
    want = oracle.roi_head_forward(feat, boxes, params, h)
    with torch.no_grad():
        bf = heads._shared_roi_transform([T.dev(feat)], [p.proposal_boxes for p in props])
        scores, deltas = heads.box_predictor(heads._pooled_mean(bf))
    print(dtype, "logits max err", np.abs(scores.cpu().numpy() - want["scores"]).max(), "max |logit|", np.abs(want["scores"]).max(),
          "res5 rel err", np.abs(bf.cpu().numpy() - want["res5"]).max() / np.abs(want["res5"]).max())
