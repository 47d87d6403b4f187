"""Text-bank producer and on-disk format: JSON {class: [D floats]} -> class_emb_mtx
[K+1, D] with a zero background row (coco_instances.py:228-254), BERT token pooling
(tools/coco_bert_embeddings.py:26-30), per-dataset installation (trainer.py:365-396)."""
import json

import numpy as np
import pytest
import torch

from locov_amd import text_bank
from locov_amd.config import get_cfg
from locov_amd.roi_heads import build_box_predictor


def test_json_to_class_emb_mtx(tmp_path):
    rng = np.random.default_rng(0)
    classes = ["person", "bicycle", "hot dog", "umbrella"]
    emb = {c: rng.standard_normal(768).astype(np.float32).tolist() for c in classes + ["unused class"]}
    p = tmp_path / "coco_nouns_bertemb.json"
    p.write_text(json.dumps(emb))
    loaded = text_bank.load_noun_embeddings(str(p))
    order = ["hot dog", "person", "umbrella"]                    # a dataset's thing_classes order
    mtx = text_bank.build_class_emb_mtx(loaded, order)
    assert mtx.shape == (4, 768) and mtx.dtype == np.float32
    assert np.all(mtx[-1] == 0)                                   # background row
    for i, c in enumerate(order):
        np.testing.assert_array_equal(mtx[i], np.asarray(emb[c], np.float32))
    with pytest.raises(KeyError):
        text_bank.build_class_emb_mtx(loaded, ["zebra"])


def test_token_pooling_matches_reference_formula():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(5, 7, 16, generator=g)
    special = torch.zeros(5, 7, dtype=torch.int64)
    special[:, 0] = 1
    special[:, 4:] = 1
    special[2, 3] = 1
    mask = (1 - special).to(torch.float32)
    want = (x * mask[:, :, None]).sum(1) / mask.sum(1)[:, None]          # coco_bert_embeddings.py:26-30
    got = text_bank.pool_token_embeddings(x.numpy(), special.numpy())
    np.testing.assert_allclose(got, want.numpy(), rtol=1e-6, atol=1e-7)


def test_bank_swap_like_evaluation(tmp_path):
    """Evaluation swaps the 48 / 17 / 65-class banks in and out (trainer.py:187-191,254-257)."""
    cfg = get_cfg()
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    bp = build_box_predictor(cfg, 64)
    rng = np.random.default_rng(2)
    cache = text_bank.TextBankCache(device="cpu")
    for name, k in (("coco_seen", 48), ("coco_unseen", 17), ("coco_all", 65)):
        m = np.zeros((k + 1, 768), np.float32)
        m[:k] = rng.standard_normal((k, 768)) * 0.05
        cache.add(name, m)
    with pytest.raises(ValueError):
        cache.add("bad", np.ones((4, 768), np.float32))              # no zero background row

    class Heads:
        num_classes = None
    heads = Heads()
    for name, k in (("coco_unseen", 17), ("coco_all", 65), ("coco_seen", 48)):
        cache.install(name, bp, heads)
        assert bp.num_classes == k and heads.num_classes == k
        assert tuple(bp.cls_score.weight.shape) == (k + 1, 768)
        assert torch.equal(bp.cls_score.weight, cache.get(name))
    assert cache.names() == ["coco_seen", "coco_unseen", "coco_all"]
