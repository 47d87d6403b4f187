"""GPU parity of the multi-token grounding predictor: locov_amd GroundingModule /
EmbeddingGroundingFastRCNNOutputLayers against vectors recorded from the reference's own
GroundingModule (tests/golden/g6_grounding_module.npz) and against the oracle at a larger size."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G6 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g6_grounding_module.npz"))


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device")
    import locov_amd
    from locov_amd import _lib
    _lib.load()
    return locov_amd


@pytest.mark.parametrize("metric", ["dot", "cosine"])
@pytest.mark.parametrize("align", ["softmax", "hardmax"])
@pytest.mark.parametrize("temp", [1, 10])
def test_grounding_module_matches_reference_vectors(pkg, metric, align, temp):
    from locov_amd import ops
    from locov_amd.roi_heads import GroundingModule
    ntok = G6["ntok"]
    gm = GroundingModule(48, len(ntok), 5, local_metric=metric, alignment=align, temperature=float(temp),
                         normalize_emb=metric == "cosine")
    gm.set_class_embeddings({k: torch.from_numpy(G6[f"emb{k}"]) for k in range(len(ntok))}, "cuda")
    x = torch.from_numpy(G6["image_emb"]).cuda()
    if metric == "cosine":
        x = ops.rownorm(x, ops.NORM_L2)
    tag = f"{metric}_{align}_t{temp}"
    for _ in range(2):                                        # calling twice changes nothing
        scores, att = gm(x)
        np.testing.assert_allclose(scores.cpu().numpy(), G6[tag + "_scores"], atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(att.cpu().numpy(), G6[tag + "_att"], atol=2e-6)
    assert tuple(gm.mask_emb.shape) == (len(ntok) + 1, 5) and int(gm.num_tok[-1]) == 0


def test_predictor_with_lvis_size_token_bank_vs_oracle(pkg, oracle):
    """EmbeddingGroundingFastRCNNOutputLayers built from a config; 1203 classes of 1-4 tokens."""
    cfg = pkg.config.get_cfg()
    cfg.MODEL.ROI_BOX_HEAD.NAME = "EmbeddingGroundingFastRCNNOutputLayers"
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    cfg.MODEL.ROI_BOX_HEAD.EMB_DIM = 64
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = 1203
    pred = pkg.build_box_predictor(cfg, 256).cuda().eval()
    rng = np.random.default_rng(4)
    ntok = rng.integers(1, 5, size=1203)
    embs = {k: (rng.standard_normal((n, 64)) * 0.2).astype(np.float32) for k, n in enumerate(ntok)}
    pred.set_class_embeddings({k: torch.from_numpy(v) for k, v in embs.items()})
    assert pred.num_classes == 1203
    x = np.maximum(rng.standard_normal((300, 256)), 0).astype(np.float32)
    with torch.no_grad():
        scores, deltas = pred(torch.from_numpy(x).cuda())
    emb = oracle.linear(x, pred.emb_pred.weight.detach().cpu().numpy(), pred.emb_pred.bias.detach().cpu().numpy())
    want, _ = oracle.grounding_module_forward(emb, [embs[k] for k in range(1203)], temperature=10.0)
    assert tuple(scores.shape) == (300, 1204) and tuple(deltas.shape) == (300, 4)
    np.testing.assert_allclose(scores.cpu().numpy(), want, atol=1e-4)
    assert torch.all(scores[:, -1] == 0)
