"""Pins the CPU oracle against vectors produced by the reference's own code
(tests/golden/make_golden.py)."""
import os

import numpy as np


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_g1_normalize_vec(oracle, golden_dir):
    g = _load(golden_dir, "g1_rownorm.npz")
    got = oracle.normalize_vec(g["x"])
    np.testing.assert_allclose(got, g["normalize_vec"], rtol=1e-6, atol=1e-7)
    assert np.all(got[7] == 0)                      # zero row stays zero (eps clamp)
    assert np.all(np.isfinite(got))


def test_g1_standardize_vec(oracle, golden_dir):
    g = _load(golden_dir, "g1_rownorm.npz")
    got = oracle.standardize_vec(g["x"])
    np.testing.assert_allclose(got, g["standardize_vec"], rtol=2e-5, atol=2e-6)
    assert np.all(got[7] == 0) and np.all(got[13] == 0)   # 0/(0+1e-12) = 0


def test_g1_l2_normalize_matches_on_nonzero_rows(oracle, golden_dir):
    # ovr/misc.py:l2_normalize differs from F.normalize only for ||x|| < eps
    g = _load(golden_dir, "g1_rownorm.npz")
    got = oracle.normalize_vec(g["x"])
    rows = [i for i in range(g["x"].shape[0]) if i not in (7, 11)]
    np.testing.assert_allclose(got[rows], g["l2_normalize"][rows], rtol=1e-6, atol=1e-7)


def test_g2_dot_similarity(oracle, golden_dir):
    g = _load(golden_dir, "g2_dot_similarity.npz")
    np.testing.assert_allclose(oracle.linear(g["emb"], g["bank81"], None), g["sim81"], atol=1e-5)
    np.testing.assert_allclose(oracle.linear(g["emb96"], g["bank1204"], None), g["sim1204"], atol=1e-5)
    assert np.all(g["sim81"][:, -1] == 0)            # zero background row -> zero logit


def test_g3_box_predictor(oracle, golden_dir):
    g = _load(golden_dir, "g3_box_predictor.npz")
    for tag, norm, std in (("dot", False, False), ("norm", True, False), ("std", False, True)):
        cls_w, cls_b, k = oracle.set_class_embeddings(g["bank"], norm, std)
        assert k == 80
        np.testing.assert_allclose(cls_w, g[f"cls_w_{tag}"], rtol=2e-5, atol=2e-6)
        assert np.all(g[f"cls_b_{tag}"] == 0)
        scores, deltas, _ = oracle.box_predictor_forward(
            g["feats"], g["emb_w"], g["emb_b"], g["bbox_w"], g["bbox_b"], cls_w, cls_b, norm, std)
        np.testing.assert_allclose(deltas, g[f"deltas_{tag}"], atol=1e-6)
        np.testing.assert_allclose(scores, g[f"scores_{tag}"], atol=1e-4, rtol=1e-5)
        assert np.all(scores[:, -1] == 0)


def test_g4_grounding_head(oracle, golden_dir):
    """GroundingHead.forward restatement vs the reference's own outputs (B = 1, 2, 4; ragged masks)."""
    g = _load(golden_dir, "g4_grounding_head.npz")
    for B in (1, 2, 4):
        p = f"b{B}_"
        losses, info, w2r, r2w = oracle.grounding_head_forward(
            g[p + "region_features"], g[p + "region_mask"], g[p + "input_embeddings"], g[p + "attention_mask"],
            g[p + "special_tokens_mask"], g["v2l_w"], g["v2l_b"], temperature=10.0)
        np.testing.assert_allclose(w2r, g[p + "w2r"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(r2w, g[p + "r2w"], rtol=2e-5, atol=2e-6)
        for name, want in zip(g[p + "loss_names"], g[p + "losses"]):
            assert abs(losses[str(name)] - float(want)) <= 2e-5, (B, name)
        for name, want in zip(g[p + "info_names"], g[p + "info"]):
            assert info[str(name)] == float(want), (B, name)


def _g6_cases():
    return [(m, a, t) for m in ("dot", "cosine") for a in ("softmax", "hardmax") for t in (1, 10)]


def test_g6_grounding_module(oracle, golden_dir):
    """GroundingModule.forward restatement vs the reference's own outputs (multi-token classes, both
    metrics and alignments, token-less background row)."""
    g = _load(golden_dir, "g6_grounding_module.npz")
    embs = [g[f"emb{k}"] for k in range(len(g["ntok"]))]
    for metric, align, temp in _g6_cases():
        cos = metric == "cosine"
        x = oracle.normalize_vec(g["image_emb"]) if cos else g["image_emb"]
        toks = [oracle.normalize_vec(e) for e in embs] if cos else embs
        scores, att = oracle.grounding_module_forward(x, toks, float(temp), cosine=cos, hardmax=align == "hardmax")
        tag = f"{metric}_{align}_t{temp}"
        np.testing.assert_allclose(scores, g[tag + "_scores"], atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(att, g[tag + "_att"], atol=2e-6)
        assert np.all(scores[:, -1] == 0) and np.all(att[:, -1] == 0)        # background row
