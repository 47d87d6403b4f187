"""The fused detection post-processing (csrc/detect.hip, ops.detect_postprocess) against the torch-op chain it replaces
(locov_amd/roi_heads/box_emb_head.py: predict_boxes + predict_probs + fast_rcnn_inference, itself gated against the oracle's
pipeline in tests/test_gpu_stt.py / test_gpu_edge_cases.py): SURVEY.md 8a-10, ovr/modeling/roi_heads/roi_emb_heads.py:280,357.
The detections must be BIT-IDENTICAL: same candidates, same NMS decisions on the shifted boxes, same order, same tie-breaking."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device")
    import locov_amd
    from locov_amd import _lib
    _lib.load()
    return locov_amd


def _predictor(pkg, classes, topk=100, thresh=0.05):
    cfg = pkg.config.get_cfg()
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = classes
    cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = thresh
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    cfg.TEST.DETECTIONS_PER_IMAGE = topk
    pred = pkg.roi_heads.box_emb_head.build_box_predictor(cfg, 256).cuda().eval()
    return pred


def _inputs(pkg, sizes, classes, sigma, seed, image_shapes=None, dup_rows=0, crowd=0):
    from locov_amd.structures import Boxes, Instances
    g = torch.Generator().manual_seed(seed)
    R = sum(sizes)
    logits = torch.randn(R, classes + 1, generator=g) * sigma
    logits[:, -1] = 0.0
    deltas = torch.randn(R, 4, generator=g) * torch.tensor([1.0, 1.0, 0.5, 0.5])
    props, r0 = [], 0
    image_shapes = image_shapes or [(800, 1333)] * len(sizes)
    boxes_all = []
    for n, (h, w) in zip(sizes, image_shapes):
        xy = torch.rand(n, 2, generator=g) * torch.tensor([w * 0.8, h * 0.8])
        wh = torch.rand(n, 2, generator=g) * torch.tensor([w * 0.4, h * 0.4]) + 8.0
        b = torch.cat([xy, xy + wh], dim=1)
        if crowd and n:                       # many proposals on a few objects: long same-class NMS chains
            centres = b[:crowd].clone()
            b = centres[torch.arange(n) % crowd] + torch.randn(n, 4, generator=g) * 6.0
            b[:, 2:] = torch.maximum(b[:, 2:], b[:, :2] + 4.0)
        boxes_all.append(b)
        r0 += n
    if dup_rows and R > 2 * dup_rows:         # identical rows: equal scores -> the tie-breaking rules decide
        logits[dup_rows:2 * dup_rows] = logits[:dup_rows]
        deltas[dup_rows:2 * dup_rows] = deltas[:dup_rows]
        flat = torch.cat(boxes_all)
        flat[dup_rows:2 * dup_rows] = flat[:dup_rows]
        boxes_all = list(torch.split(flat, sizes))
    for n, shape, b in zip(sizes, image_shapes, boxes_all):
        p = Instances(shape)
        p.proposal_boxes = Boxes(b.float().cuda())
        props.append(p)
    return (logits.cuda(), deltas.cuda()), props


def _run(pkg, pred, predictions, props, fused, monkeypatch):
    beh = pkg.roi_heads.box_emb_head
    monkeypatch.setattr(beh, "_FUSED_POSTPROCESS", fused)
    with torch.no_grad():
        res, kept = pred.inference(predictions, props)
    torch.cuda.synchronize()
    return res, kept


def _same(a, b):
    ra, ka = a
    rb, kb = b
    assert len(ra) == len(rb)
    for x, y, kx, ky in zip(ra, rb, ka, kb):
        assert len(x) == len(y), (len(x), len(y))
        assert x.image_size == y.image_size
        assert torch.equal(x.pred_classes, y.pred_classes)
        assert torch.equal(x.scores, y.scores)
        assert torch.equal(x.pred_boxes.tensor, y.pred_boxes.tensor)
        assert torch.equal(kx, ky)


@pytest.mark.parametrize("sizes,classes,sigma,kw", [
    ([1000], 1203, 3.0, {}),                                                   # the evaluation call: one image (coco_stt.yaml:50)
    ([1000] * 8, 1203, 3.0, {}),                                               # the bench's batch
    ([300, 0, 1, 517], 80, 2.0, {"image_shapes": [(800, 1333), (640, 960), (480, 640), (1067, 800)]}),   # ragged, an empty image
    ([1000], 48, 1.5, {"crowd": 12}),                                          # few classes: hundreds of candidates per class
    ([400, 400], 65, 2.5, {"dup_rows": 50}),                                   # exact score ties
    ([1500], 3, 0.2, {"crowd": 40}),                                           # every proposal a candidate of every class: 1 500 per class
    ([2000, 700], 2, 0.3, {}),                                                 # ... and spread-out boxes (most survive: long wait chains)
    ([2700], 3, 0.2, {}),                                                      # 8 100 candidates: the 8 192-slot sort, full
    ([37] * 64, 20, 1.0, {}),                                                  # the most images one call takes
    ([64], 5, 0.1, {}),                                                        # nearly uniform scores: everything passes 0.05 at 6 columns
    ([200], 1203, 0.01, {}),                                                   # nothing passes the threshold
])
def test_fused_postprocess_is_bit_identical_to_the_torch_chain(pkg, monkeypatch, sizes, classes, sigma, kw):
    pred = _predictor(pkg, classes)
    predictions, props = _inputs(pkg, sizes, classes, sigma, seed=len(sizes) * 7 + classes, **kw)
    want = _run(pkg, pred, predictions, props, False, monkeypatch)
    got = _run(pkg, pred, predictions, props, True, monkeypatch)
    _same(got, want)
    if sigma >= 1.0:
        assert max(len(r) for r in got[0]) > 0
    if sigma == 0.01:
        assert all(len(r) == 0 for r in got[0])


@pytest.fixture(scope="module")
def oracle_mod():
    from oracle import lsm_oracle
    lsm_oracle.build()
    return lsm_oracle


@pytest.mark.parametrize("sizes,classes,sigma,shapes", [([400, 250], 80, 2.0, [(800, 1333), (640, 960)]),
                                                        ([1000], 1203, 3.0, [(800, 1333)])],        # BASELINE's evaluation call: full size
                         ids=["two_images_80_classes", "evaluation_call_1000x1203"])
def test_fused_postprocess_vs_the_oracle_pipeline(pkg, oracle_mod, monkeypatch, sizes, classes, sigma, shapes):
    """The device pipeline against the CPU oracle's apply_deltas -> softmax -> fast_rcnn_inference_single_image on the same logits,
    deltas and proposals (numpy arithmetic: a decoded coordinate may differ in its last bits, so a borderline NMS decision may too):
    same number of detections, classes and scores agree as the end-to-end gate of tests/test_gpu_stt.py asks."""
    pred = _predictor(pkg, classes)
    predictions, props = _inputs(pkg, sizes, classes, sigma, seed=19, image_shapes=shapes)
    got, _ = _run(pkg, pred, predictions, props, True, monkeypatch)
    logits, deltas = predictions[0].cpu().numpy(), predictions[1].cpu().numpy()
    r0 = 0
    for n, p, res in zip(sizes, props, got):
        boxes = oracle_mod.apply_deltas(deltas[r0:r0 + n], p.proposal_boxes.tensor.cpu().numpy())
        cb, cs, cc = oracle_mod.fast_rcnn_inference_single_image(boxes, oracle_mod.softmax(logits[r0:r0 + n]), p.image_size,
                                                                 pred.test_score_thresh, pred.test_nms_thresh, pred.test_topk_per_image)
        db, ds, dc = res.pred_boxes.tensor.cpu().numpy(), res.scores.cpu().numpy(), res.pred_classes.cpu().numpy()
        assert len(ds) == len(cs) == 100
        np.testing.assert_allclose(np.sort(ds)[::-1], np.sort(cs)[::-1], atol=1e-6)
        od, oc = np.lexsort((db[:, 0], dc, -np.round(ds, 4))), np.lexsort((cb[:, 0], cc, -np.round(cs, 4)))
        assert (dc[od] == cc[oc]).mean() > 0.98
        np.testing.assert_allclose(db[od][dc[od] == cc[oc]], cb[oc][dc[od] == cc[oc]], atol=2e-3)
        r0 += n


def test_box_decoding_and_clipping_round_as_the_torch_ops(pkg):
    """det_decode_clip_kernel == Box2BoxTransform.apply_deltas + Boxes.clip bit for bit (torch divides by the weights through a
    multiplication with the fp32 reciprocal; the clamp of dw / dh; exp)."""
    from locov_amd import ops
    beh = pkg.roi_heads.box_emb_head
    g = torch.Generator().manual_seed(4)
    R = 5000
    deltas = (torch.randn(R, 4, generator=g) * torch.tensor([3.0, 3.0, 4.0, 4.0])).cuda()       # (some dw / dh above the clamp of 4.135)
    xy = torch.rand(R, 2, generator=g) * 900
    boxes = torch.cat([xy, xy + torch.rand(R, 2, generator=g) * 600 + 1], dim=1).cuda()
    t = beh.Box2BoxTransform((10.0, 10.0, 5.0, 5.0))
    want = t.apply_deltas(deltas, boxes)
    want = torch.stack([want[:, 0].clamp(0, 1333), want[:, 1].clamp(0, 800), want[:, 2].clamp(0, 1333), want[:, 3].clamp(0, 800)], dim=1)
    # every proposal its own class with probability 1 -> every box comes back, in row order (equal scores: candidate order)
    probs = torch.zeros(R, R + 1, device="cuda")
    probs[torch.arange(R), torch.arange(R)] = 1.0
    out = ops.detect_postprocess(probs, deltas, boxes, [R], [(800, 1333)], t.weights, t.scale_clamp, 0.05, 0.5, R)
    assert out is not None
    got_boxes, scores, classes, rows, counts = out
    assert counts == [R] and torch.equal(rows[0], torch.arange(R, device="cuda")) and torch.equal(classes[0], rows[0])
    assert torch.equal(got_boxes[0], want)


def test_fused_postprocess_hands_over_what_it_does_not_take(pkg, monkeypatch):
    """Non-finite predictions (the reference drops such proposals with a warning) and more candidates per image than the LDS sort
    holds are flagged on the device: inference() then runs the torch chain -- same results as with the fused path switched off."""
    import warnings
    from locov_amd import ops
    pred = _predictor(pkg, 80)
    predictions, props = _inputs(pkg, [300], 80, 2.0, seed=3)
    predictions[1][7, 0] = float("inf")              # (an infinite dw alone is clamped to a finite box; dx is not)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        want = _run(pkg, pred, predictions, props, False, monkeypatch)
        got = _run(pkg, pred, predictions, props, True, monkeypatch)
    _same(got, want)
    probs = torch.softmax(predictions[0], dim=-1)
    assert ops.detect_postprocess(probs, predictions[1], props[0].proposal_boxes.tensor, [300], [(800, 1333)], (10.0, 10.0, 5.0, 5.0),
                                  4.135, 0.05, 0.5, 100) is None
    # 300 rows x 80 classes at a threshold of 0: 24 000 candidates > 8 192
    pred0 = _predictor(pkg, 80, thresh=0.0)
    predictions, props = _inputs(pkg, [300], 80, 2.0, seed=5)
    want = _run(pkg, pred0, predictions, props, False, monkeypatch)
    got = _run(pkg, pred0, predictions, props, True, monkeypatch)
    _same(got, want)
    assert ops.detect_postprocess(torch.softmax(predictions[0], dim=-1), predictions[1], props[0].proposal_boxes.tensor, [300], [(800, 1333)],
                                  (10.0, 10.0, 5.0, 5.0), 4.135, 0.0, 0.5, 100) is None


def test_evaluation_call_through_the_heads_uses_the_fused_path(pkg, monkeypatch):
    """roi_heads(images, features, proposals, None) (roi_emb_heads.py:351-360) with the fused post-processing on / off: identical
    Instances; and the fused call makes ONE host read behind the predictor (counted through torch's sync-debug mode)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    args = bench.parse(["--images", "2", "--proposals", "300", "--classes", "80", "--no-cpu-baseline"])
    wl = bench.Workload(args, torch.device("cuda", 0))
    beh = pkg.roi_heads.box_emb_head
    outs = {}
    for fused in (False, True):
        monkeypatch.setattr(beh, "_FUSED_POSTPROCESS", fused)
        inst, _ = wl.step_eval(2)
        torch.cuda.synchronize()
        outs[fused] = inst
    assert sum(len(x) for x in outs[True]) > 0
    for a, b in zip(outs[True], outs[False]):
        assert len(a) == len(b) and torch.equal(a.scores, b.scores) and torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor)
        assert torch.equal(a.pred_classes, b.pred_classes)
