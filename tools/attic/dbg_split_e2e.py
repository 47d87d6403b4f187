"""Whole head vs the CPU oracle with the Res5 GEMMs on the f32 MFMA and in split-operand arithmetic, at several input
magnitudes (the split form's operand scales assume activations between ~1e-2 and 4e3)."""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import locov_amd as pkg
from oracle import lsm_oracle as oracle
import test_gpu_roi_heads as T
oracle.build()
for fscale in (1.0, 30.0, 0.03):
    for dtype in ["fp32", "f16x2"]:
        cfg = T._small_cfg(pkg)
        cfg.MODEL.ROI_BOX_HEAD.RES5_DTYPE = dtype
        heads, params, h = T._make_heads(pkg, oracle, cfg, 80, 1992)
        rng = np.random.default_rng(1992)
        feat = (np.maximum(rng.standard_normal((2, 128, 50, 84)), 0) * fscale).astype(np.float32)
        feat[:, :, ::9, ::7] *= 8.0                                # heavy tail
        props, boxes = T._proposals(pkg, oracle, rng, 2, 60)
        want = oracle.roi_head_forward(feat, boxes, params, h)
        with torch.no_grad():
            bf = heads._shared_roi_transform([T.dev(feat)], [p.proposal_boxes for p in props])
            scores, deltas = heads.box_predictor(heads._pooled_mean(bf))
        m = np.abs(want["scores"]).max()
        print(f"feat x{fscale:<5} {dtype:6s} logits max err {np.abs(scores.cpu().numpy() - want['scores']).max():.3e} (max |logit| {m:.3g}, rel {np.abs(scores.cpu().numpy() - want['scores']).max() / m:.2e})"
              f"   res5 rel err {np.abs(bf.cpu().numpy() - want['res5']).max() / np.abs(want['res5']).max():.2e}  max |res5| {np.abs(want['res5']).max():.3g}")
