"""The seam to Detectron2's own objects (`register_with_detectron2(override=True)` + the reference's yaml
names), exercised against tests/stubs/detectron2 -- a restatement of the Detectron2 public API the path touches, with
classes DISTINCT from locov_amd.structures' (Detectron2 itself cannot be installed here):
  * the registry drop-in: ovr/__init__.py:9-10 registers the reference's heads, ours replace them under the same names
    and detectron2.modeling.roi_heads.build_roi_heads(cfg, shape) then builds the MI355X implementation;
  * training: label_and_sample_proposals on D2-typed Instances / Boxes (roi_emb_heads.py:25-118, add_ground_truth_to_proposals
    mixes gt_boxes into the proposals);
  * inference: the results are D2 Instances with D2 Boxes, so that the meta-architectures' detector_postprocess
    (ovr_rcnn.py:111, distill_prop_mmss_gcnn.py:556: pred_boxes.scale / .clip / .nonempty) works on them."""
import os
import sys

import numpy as np
import pytest
import torch

STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")


@pytest.fixture()
def d2():
    """Makes the stub importable as `detectron2` for one test and removes every trace afterwards."""
    assert not any(m == "detectron2" or m.startswith("detectron2.") for m in sys.modules), "a real detectron2 is loaded"
    sys.path.insert(0, STUBS)
    import detectron2
    import detectron2.modeling.postprocessing
    import detectron2.modeling.roi_heads
    import detectron2.structures
    import detectron2.utils.events
    yield detectron2
    sys.path.remove(STUBS)
    for m in [m for m in sys.modules if m == "detectron2" or m.startswith("detectron2.")]:
        del sys.modules[m]


def _cfg():
    from locov_amd.config import get_cfg
    cfg = get_cfg()
    cfg.MODEL.RESNETS.RES2_OUT_CHANNELS = 32
    cfg.MODEL.RESNETS.WIDTH_PER_GROUP = 8
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    cfg.MODEL.ROI_BOX_HEAD.EMB_DIM = 96
    cfg.MODEL.ROI_HEADS.NAME = "EmbeddingProposalsRes5ROIHeads"
    cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 16
    cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION = 1.0
    return cfg


def _d2_batch(d2, oracle, device, n_img=2, r=40, n_gt=3, seed=3):
    from detectron2.structures import Boxes, Instances
    rng = np.random.default_rng(seed)
    props, targets = [], []
    for _ in range(n_img):
        gt = oracle.synth_boxes(rng, n_gt)
        b = oracle.synth_boxes(rng, r)
        b[:n_gt] = gt + 1.5
        p = Instances((800, 1333))
        p.proposal_boxes = Boxes(torch.from_numpy(b).to(device))
        p.objectness_logits = torch.zeros(r, device=device)
        t = Instances((800, 1333))
        t.gt_boxes = Boxes(torch.from_numpy(gt).to(device))
        t.gt_classes = torch.from_numpy(rng.integers(0, 80, n_gt)).to(device)
        targets.append(t)
        props.append(p)
    return props, targets


def test_registry_drop_in_and_d2_typed_training_labels(d2, oracle):
    from detectron2.modeling.roi_heads import ROI_HEADS_REGISTRY, build_roi_heads
    from detectron2.structures import Boxes as D2Boxes, Instances as D2Instances
    from detectron2.utils.events import EventStorage
    import locov_amd
    from locov_amd.roi_heads import register_with_detectron2
    from locov_amd.structures import ShapeSpec

    class EmbeddingProposalsRes5ROIHeads:          # stands for the reference's class, registered by `import ovr`
        pass
    ROI_HEADS_REGISTRY.register(EmbeddingProposalsRes5ROIHeads)
    assert register_with_detectron2(override=False) is True
    assert ROI_HEADS_REGISTRY.get("EmbeddingProposalsRes5ROIHeads") is EmbeddingProposalsRes5ROIHeads     # kept without override
    assert ROI_HEADS_REGISTRY.get("EmbeddingRes5ROIHeads") is locov_amd.EmbeddingRes5ROIHeads
    assert register_with_detectron2(override=True) is True
    assert ROI_HEADS_REGISTRY.get("EmbeddingProposalsRes5ROIHeads") is locov_amd.EmbeddingProposalsRes5ROIHeads
    heads = build_roi_heads(_cfg(), {"res4": ShapeSpec(channels=128, stride=16)})          # Detectron2's own builder
    assert isinstance(heads, locov_amd.EmbeddingProposalsRes5ROIHeads) and heads.output_shape == 256

    props, targets = _d2_batch(d2, oracle, "cpu")
    heads.num_classes = 80
    with EventStorage() as storage:                  # under Detectron2's trainer the scalars go to ITS storage
        sampled = heads.label_and_sample_proposals(props, targets)
    assert set(storage.scalars) == {"roi_head/num_fg_samples", "roi_head/num_bg_samples"}
    for s, t in zip(sampled, targets):
        assert isinstance(s, D2Instances) and isinstance(s.proposal_boxes, D2Boxes) and isinstance(s.gt_boxes, D2Boxes)
        assert len(s) == 16 and s.has("gt_classes") and s.has("fg_proposal") and s.has("objectness_logits")
        fg = s.fg_proposal.bool()
        assert int(fg.sum()) >= len(t)                # the appended ground-truth boxes match themselves
        assert torch.all(s.gt_classes[~fg] == 80)


def test_inference_results_are_d2_objects_for_detector_postprocess(d2, oracle):
    """CPU tensors: the predictor's inference (box decoding, softmax, NMS, top-k -- host logic) on D2-typed proposals."""
    from detectron2.modeling.postprocessing import detector_postprocess
    from detectron2.structures import Boxes as D2Boxes, Instances as D2Instances
    from locov_amd.roi_heads import build_box_predictor
    bp = build_box_predictor(_cfg(), 64)
    rng = np.random.default_rng(5)
    bank = np.zeros((81, 96), np.float32)
    bank[:80] = rng.standard_normal((80, 96)) * 0.05
    bp.set_class_embeddings(bank)
    props, _ = _d2_batch(d2, oracle, "cpu", r=30)
    scores = torch.from_numpy(rng.standard_normal((60, 81)).astype(np.float32)) * 3
    deltas = torch.from_numpy(rng.standard_normal((60, 4)).astype(np.float32)) * 0.1
    results, kept = bp.inference((scores, deltas), props)
    assert len(results) == 2
    for r in results:
        assert isinstance(r, D2Instances) and isinstance(r.pred_boxes, D2Boxes)
        before = r.pred_boxes.tensor.clone()
        out = detector_postprocess(r, 400, 667)       # e.g. INPUT.MAX_SIZE_TEST 400 images evaluated at their original size
        assert isinstance(out, D2Instances) and out.image_size == (400, 667)
        keep = (before[:, 2] > before[:, 0]) & (before[:, 3] > before[:, 1])
        np.testing.assert_allclose(out.pred_boxes.tensor.numpy(),
                                   (before * torch.tensor([667 / 1333, 400 / 800, 667 / 1333, 400 / 800]))[keep].numpy(), rtol=1e-6)


def test_own_boxes_have_the_d2_methods_postprocessing_needs():
    from locov_amd.structures import Boxes
    b = Boxes(torch.tensor([[10.0, 20.0, 110.0, 220.0], [5.0, 5.0, 5.0, 9.0]]))
    b.scale(0.5, 0.25)
    np.testing.assert_allclose(b.tensor.numpy(), [[5.0, 5.0, 55.0, 55.0], [2.5, 1.25, 2.5, 2.25]])
    assert b.nonempty().tolist() == [True, False] and b.inside_box((60, 60)).tolist() == [True, True]


@pytest.mark.gpu
def test_roi_heads_to_detector_postprocess_on_the_gpu(d2, oracle):
    """The whole inference seam on the device: D2-typed proposals -> EmbeddingProposalsRes5ROIHeads (HIP path) ->
    D2 Instances -> detector_postprocess; and a training forward on D2-typed targets."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device")
    from detectron2.modeling.postprocessing import detector_postprocess
    from detectron2.modeling.roi_heads import build_roi_heads
    from detectron2.structures import Boxes as D2Boxes, Instances as D2Instances
    from locov_amd.roi_heads import register_with_detectron2
    from locov_amd.structures import ShapeSpec
    register_with_detectron2(override=True)
    torch.manual_seed(0)
    heads = build_roi_heads(_cfg(), {"res4": ShapeSpec(channels=128, stride=16)}).cuda().eval()
    rng = np.random.default_rng(7)
    bank = np.zeros((81, 96), np.float32)
    bank[:80] = rng.standard_normal((80, 96))
    heads.box_predictor.set_class_embeddings(bank)
    heads.num_classes = 80
    feat = torch.from_numpy(rng.standard_normal((2, 128, 50, 84)).astype(np.float32)).cuda()
    props, targets = _d2_batch(d2, oracle, "cuda")
    with torch.no_grad():
        results, losses = heads(None, {"res4": feat}, props, None)
    assert losses == {} and len(results) == 2
    for r in results:
        assert isinstance(r, D2Instances) and isinstance(r.pred_boxes, D2Boxes) and len(r) > 0
        out = detector_postprocess(r, 400, 667)
        assert float(out.pred_boxes.tensor[:, 2].max()) <= 667 and float(out.pred_boxes.tensor[:, 3].max()) <= 400
    heads.train()
    grid, box_feats, sampled, losses = heads(None, {"res4": feat.requires_grad_(True)}, props, targets)
    assert all(isinstance(s, D2Instances) and isinstance(s.proposal_boxes, D2Boxes) for s in sampled)
    losses["loss_box_reg"].backward()
    assert feat.grad is not None and torch.isfinite(feat.grad).all()
