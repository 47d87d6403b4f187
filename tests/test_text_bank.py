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


def test_g7_token_pooling_matches_the_reference_script(golden_dir):
    """pool_token_embeddings against the embeddings tools/coco_bert_embeddings.py:26-34 produced (its own statements,
    run by tests/golden/make_golden.py on a stand-in for the BERT encoder output), incl. the JSON it writes."""
    import os
    g = np.load(os.path.join(golden_dir, "g7_text_bank.npz"))
    got = text_bank.pool_token_embeddings(g["pool_input_embeddings"], g["pool_special_tokens_mask"])
    np.testing.assert_allclose(got, g["pool_embeddings"], rtol=1e-6, atol=1e-7)
    table = json.loads(str(g["pool_json"]))
    assert list(table) == [f"class {k}" for k in range(got.shape[0])]
    np.testing.assert_allclose(np.asarray([table[k] for k in table], np.float32), got, rtol=1e-6, atol=1e-7)


def test_g7_class_emb_mtx_layouts_match_the_reference(golden_dir, tmp_path):
    """COCO (coco_instances.py:237-254: a multi-token class keeps a zero row and goes to the class_embeddings dict) and
    LVIS (lvis_instances.py:269-278) bank layouts against what the reference's statements built from the same JSON."""
    import os
    g = np.load(os.path.join(golden_dir, "g7_text_bank.npz"))
    p = tmp_path / "nouns.json"
    p.write_text(str(g["bank_json"]))
    table = text_bank.load_noun_embeddings(str(p))
    mtx, per_class = text_bank.build_class_emb_mtx(table, [str(c) for c in g["bank_thing_classes"]], return_class_embeddings=True)
    np.testing.assert_array_equal(mtx, g["coco_class_emb_mtx"])
    assert sorted(k for k, v in per_class.items() if v.ndim == 2) == g["coco_multi_token_idx"].tolist()
    np.testing.assert_array_equal(per_class[2], g["coco_multi_token_emb"])
    assert np.all(mtx[2] == 0) and np.all(mtx[-1] == 0)
    np.testing.assert_array_equal(text_bank.build_class_emb_mtx(table, [str(c) for c in g["bank_thing_classes"]]), mtx)
    lv = text_bank.build_lvis_class_emb_mtx(table, [str(c) for c in g["lvis_thing_classes"]])
    np.testing.assert_array_equal(lv, g["lvis_class_emb_mtx"])
    assert text_bank.build_lvis_class_emb_mtx(None, ["a"]) is None                 # no obj_file: no bank (lvis_instances.py:262-263)
    with pytest.raises(ValueError):
        text_bank.build_lvis_class_emb_mtx(table, ["noun3"])                        # multi-token entry: broadcast error in the reference


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
