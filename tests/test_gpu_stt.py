"""BASELINE config 5 on the device: the STT stage of the reference (configs/coco_stt.yaml) re-uses the same ROI head as
`EmbeddingRes5ROIHeads` (roi_emb_heads.py:122-306) -- 48 base classes in training, emb_pred frozen (coco_stt.yaml:34-37),
per-dataset bank swaps in evaluation (48 / 17 / 65 classes, trainer.py:187-191).

* one STT fine-tune step (forward with targets + backward) on the hand-written path against the same module on the stock
  library path (torch conv2d / MIOpen autograd);
* a synthetic evaluation: detections of the HIP path and of the CPU oracle's whole pipeline (ROIAlign -> Res5 -> mean ->
  predictor -> softmax -> decode -> NMS) scored with a COCO-style AP50 over the NOVEL classes of the generalised bank; the
  bank describes the ground-truth regions (each class embedding is the region embedding of one ground-truth box), so a
  working detector finds most of them: AP50-novel well above chance (0.8) and equal on both paths."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BASE, NOVEL = 48, 17                                   # coco_instances.py: 48 seen / 17 unseen classes, 65 generalised


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device")
    import locov_amd
    from locov_amd import _lib
    _lib.load()
    return locov_amd


# sizes: "small" keeps the suite fast; "coco_stt" = configs/coco_stt.yaml itself (res5 1024 -> (512) -> 2048, D = 768,
# 3 images x 512 sampled proposals: IMS_PER_BATCH 24 / 8 GPUs, Detectron2's default BATCH_SIZE_PER_IMAGE)
SIZES = {"small": dict(res2=32, width=8, dim=96, batch=32, mid=64),
         "coco_stt": dict(res2=256, width=64, dim=768, batch=512, mid=512)}


def _stt_heads(pkg, oracle, backend, dtype, num_classes, train, size="small"):
    from locov_amd.structures import ShapeSpec
    sz = SIZES[size]
    cfg = pkg.config.get_cfg()
    cfg.MODEL.RESNETS.RES2_OUT_CHANNELS = sz["res2"]    # res5: 128 -> (64) -> 256  /  1024 -> (512) -> 2048
    cfg.MODEL.RESNETS.WIDTH_PER_GROUP = sz["width"]
    cfg.MODEL.ROI_HEADS.NAME = "EmbeddingRes5ROIHeads"                      # coco_stt.yaml:18
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = num_classes                          # :20
    cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION = 1.0                            # :25
    cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = sz["batch"]
    cfg.MODEL.ROI_BOX_HEAD.NAME = "EmbeddingFastRCNNOutputLayers"          # :27
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True                    # :29
    cfg.MODEL.ROI_BOX_HEAD.EMB_DIM = sz["dim"]                             # (768 in the config)
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True                          # :33
    cfg.MODEL.ROI_BOX_HEAD.FREEZE_EMB_PRED = True                          # :36
    cfg.MODEL.ROI_BOX_HEAD.RES5_BACKEND = backend
    cfg.MODEL.ROI_BOX_HEAD.RES5_DTYPE = dtype
    c_in = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS * 4
    torch.manual_seed(3)
    heads = pkg.build_roi_heads(cfg, {"res4": ShapeSpec(channels=c_in, stride=16)})
    params = oracle.make_res5_params(9, in_ch=c_in, mid=sz["mid"], out_ch=heads.output_shape)
    heads.res5.load_state_dict(params)
    head = oracle.synth_head(np.random.default_rng(9), heads.output_shape, sz["dim"], num_classes)
    with torch.no_grad():
        heads.box_predictor.emb_pred.weight.copy_(torch.from_numpy(head["emb_w"]))
        heads.box_predictor.bbox_pred.weight.copy_(torch.from_numpy(head["bbox_w"]))
        heads.box_predictor.bbox_pred.bias.zero_()
    heads = heads.cuda().train(train)
    heads.box_predictor.set_class_embeddings(head["cls_w"])
    heads.num_classes = heads.box_predictor.num_classes
    return heads, c_in, params, head


def _batch(pkg, oracle, n_img, r, n_gt, seed, num_classes):
    from locov_amd.structures import Boxes, Instances
    rng = np.random.default_rng(seed)
    props, targets = [], []
    for _ in range(n_img):
        gt = oracle.synth_boxes(rng, n_gt)
        gt[:, 2:] = np.maximum(gt[:, 2:], gt[:, :2] + 24.0)
        b = oracle.synth_boxes(rng, r)
        b[:n_gt] = gt + rng.uniform(-4, 4, gt.shape).astype(np.float32)
        b[:, 2:] = np.maximum(b[:, 2:], b[:, :2] + 1.0)
        p = Instances((800, 1333))
        p.proposal_boxes = Boxes(torch.from_numpy(b).cuda())
        p.objectness_logits = torch.zeros(r, device="cuda")
        t = Instances((800, 1333))
        t.gt_boxes = Boxes(torch.from_numpy(gt).cuda())
        t.gt_classes = torch.from_numpy(rng.integers(0, num_classes, n_gt)).cuda()
        props.append(p)
        targets.append(t)
    return props, targets


@pytest.mark.parametrize("dtype,size,n_props", [("fp32", "small", 80), ("f16x2", "small", 80), ("f16x2", "coco_stt", 1000)])
def test_stt_finetune_step_matches_the_stock_library_path(pkg, oracle, dtype, size, n_props):
    """EmbeddingRes5ROIHeads.forward in training (roi_emb_heads.py:247-278): label + sample, ROIAlign + Res5 + mean,
    predictor, losses -- loss_cls is live here (no DETACH_CLASS_PREDICTOR in coco_stt.yaml), emb_pred is frozen, Res5 and
    bbox_pred train and res4 receives its gradient (BACKBONE.FREEZE_AT 2)."""
    outs = {}
    for backend in ("miopen", "hip"):
        heads, c_in, _, _ = _stt_heads(pkg, oracle, backend, dtype, BASE, train=True, size=size)
        feat = torch.randn(3, c_in, 50, 84, generator=torch.Generator().manual_seed(5)).cuda().requires_grad_(True)
        props, targets = _batch(pkg, oracle, 3, n_props, 6, seed=17, num_classes=BASE)  # 24 img / 8 GPUs (coco_stt.yaml:41)
        torch.manual_seed(77)
        out, losses = heads(None, {"res4": feat}, props, targets)
        assert out == [] and set(losses) == {"loss_cls", "loss_box_reg"}
        if size == "coco_stt":
            assert heads.batch_size_per_image == 512 and heads.output_shape == 2048 and heads.box_predictor.emb_dim == 768
        (losses["loss_cls"] + losses["loss_box_reg"]).backward()
        grads = {k: p.grad.clone() for k, p in heads.named_parameters() if p.grad is not None}
        assert not any(k.startswith("box_predictor.emb_pred") for k in grads)          # frozen (coco_stt.yaml:36)
        assert any(k.startswith("res5.") for k in grads) and "box_predictor.bbox_pred.weight" in grads
        outs[backend] = (float(losses["loss_cls"].detach()), float(losses["loss_box_reg"].detach()), feat.grad.clone(), grads)
    cm, bm, fm, pm = outs["miopen"]
    ch, bh, fh, ph = outs["hip"]
    assert abs(ch - cm) <= 2e-4 * max(abs(cm), 1e-3) and abs(bh - bm) <= 2e-4 * max(abs(bm), 1e-3)
    rel_l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    assert float(fm.abs().max()) > 0 and rel_l2(fh, fm) < 5e-3
    assert set(ph) == set(pm)
    worst = max((rel_l2(ph[k], pm[k]), k) for k in ph)
    assert worst[0] < 5e-3, worst


def _iou(a, b):
    x0, y0 = np.maximum(a[:, None, 0], b[None, :, 0]), np.maximum(a[:, None, 1], b[None, :, 1])
    x1, y1 = np.minimum(a[:, None, 2], b[None, :, 2]), np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x1 - x0, 0, None) * np.clip(y1 - y0, 0, None)
    area = lambda t: (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])
    return inter / (area(a)[:, None] + area(b)[None, :] - inter)


def ap50(dets, gts, class_ids):
    """COCO-style AP at IoU 0.5 (101-point interpolated precision), averaged over `class_ids` that have ground truth.
    dets / gts: per image (boxes [n,4], scores [n], classes [n]) / (boxes [m,4], classes [m])."""
    aps = []
    for c in class_ids:
        rows, n_gt = [], 0
        for img, ((db, ds, dc), (gb, gc)) in enumerate(zip(dets, gts)):
            g = gb[gc == c]
            n_gt += len(g)
            sel = np.nonzero(dc == c)[0]
            order = sel[np.argsort(-ds[sel], kind="stable")]
            taken = np.zeros(len(g), bool)
            for i in order:
                tp = False
                if len(g):
                    iou = _iou(db[i:i + 1], g)[0]
                    iou[taken] = -1
                    j = int(iou.argmax())
                    if iou[j] >= 0.5:
                        taken[j] = True
                        tp = True
                rows.append((ds[i], tp))
        if n_gt == 0:
            continue
        rows.sort(key=lambda r: -r[0])
        tp = np.cumsum([r[1] for r in rows]) if rows else np.zeros(0)
        fp = np.cumsum([not r[1] for r in rows]) if rows else np.zeros(0)
        rec = tp / n_gt
        prec = tp / np.maximum(tp + fp, 1)
        for i in range(len(prec) - 2, -1, -1):
            prec[i] = max(prec[i], prec[i + 1])
        q = np.linspace(0, 1, 101)
        idx = np.searchsorted(rec, q, side="left")
        aps.append(float(np.mean([prec[i] if i < len(prec) else 0.0 for i in idx])))
    return float(np.mean(aps)) if aps else float("nan")


@pytest.mark.parametrize("size", ["small", "coco_stt"])
def test_stt_synthetic_eval_ap50_novel(pkg, oracle, size):
    """Evaluation branch of EmbeddingRes5ROIHeads.forward (roi_emb_heads.py:258-262,280-282) after a bank swap to the
    generalised 65-class bank (trainer.py:187-191): AP50 over the 17 novel classes on a synthetic set whose class
    embeddings are the region embeddings of its ground-truth boxes."""
    from locov_amd.structures import Boxes, Instances
    K = BASE + NOVEL
    heads, c_in, params, head = _stt_heads(pkg, oracle, "hip", "f16x2", BASE, train=False, size=size)
    D = SIZES[size]["dim"]
    rng = np.random.default_rng(23)
    n_img, n_gt, r = 3, 8, 150
    feats = rng.standard_normal((n_img, c_in, 50, 84)).astype(np.float32)
    gts, box_lists = [], []
    novel = list(range(BASE, K))
    rng.shuffle(novel)
    used = iter(novel)
    base_pool = iter(rng.permutation(BASE))
    for i in range(n_img):
        gb = oracle.synth_boxes(rng, n_gt)
        gb[:, 2:] = np.maximum(gb[:, 2:], gb[:, :2] + 48.0)
        gc = np.array([next(used) if j < 5 else int(next(base_pool)) for j in range(n_gt)], np.int64)   # 5 novel + 3 base per image
        gts.append((gb, gc))
        pb = oracle.synth_boxes(rng, r)
        pb[:n_gt] = gb                                                   # the proposals contain the objects
        box_lists.append(pb)
    # the bank: region embeddings of a randomly initialised head are all but parallel (cosine ~0.99: ReLU features share
    # a large mean), so a class is described by how its ground-truth region DEVIATES from the mean embedding m, in the
    # subspace orthogonal to m: logit(region r, class c) = s d_r . d_c, scaled so that a region's own class sits near 12;
    # classes without an object keep a zero row (logit 0, like the background)
    embs = np.concatenate([oracle.roi_head_forward(feats[i:i + 1], [gts[i][0]], params, head)["emb"] for i in range(n_img)])
    m = embs.mean(0)
    mh = m / np.linalg.norm(m)
    dev = embs - m
    dev -= (dev @ mh)[:, None] * mh
    bank = np.zeros((K + 1, D), np.float32)
    bank[np.concatenate([g[1] for g in gts])] = (12.0 / np.mean((dev * dev).sum(1))) * dev
    heads.box_predictor.set_class_embeddings(torch.from_numpy(bank))     # the per-dataset swap (trainer.py:187-191)
    heads.num_classes = heads.box_predictor.num_classes
    assert heads.box_predictor.num_classes == K

    # device
    props = []
    for pb in box_lists:
        p = Instances((800, 1333))
        p.proposal_boxes = Boxes(torch.from_numpy(pb).cuda())
        p.objectness_logits = torch.zeros(len(pb), device="cuda")
        props.append(p)
    with torch.no_grad():
        inst, _ = heads(None, {"res4": torch.from_numpy(feats).cuda()}, props, None)
    dets_dev = [(x.pred_boxes.tensor.cpu().numpy(), x.scores.cpu().numpy(), x.pred_classes.cpu().numpy()) for x in inst]

    # the CPU oracle's whole pipeline
    head_k = dict(head, cls_w=bank)
    pred = heads.box_predictor
    dets_cpu = []
    for i in range(n_img):
        o = oracle.roi_head_forward(feats[i:i + 1], [box_lists[i]], params, head_k)
        boxes = oracle.apply_deltas(o["deltas"], box_lists[i])
        dets_cpu.append(oracle.fast_rcnn_inference_single_image(boxes, oracle.softmax(o["scores"]), (800, 1333),
                                                                pred.test_score_thresh, pred.test_nms_thresh,
                                                                pred.test_topk_per_image))
    ap_dev = {name: ap50(dets_dev, gts, ids) for name, ids in (("novel", range(BASE, K)), ("base", range(BASE)), ("all", range(K)))}
    ap_cpu = {name: ap50(dets_cpu, gts, ids) for name, ids in (("novel", range(BASE, K)), ("base", range(BASE)), ("all", range(K)))}
    assert ap_cpu["novel"] > 0.6 and ap_cpu["base"] > 0.6, ap_cpu        # the synthetic set is detectable (0.80 / 0.89 here)
    for k in ap_cpu:
        assert abs(ap_dev[k] - ap_cpu[k]) <= 1e-6, (k, ap_dev, ap_cpu)
    # and the detections themselves agree (same count, classes, scores within the logits gate)
    for (db, ds, dc), (cb, cs, cc) in zip(dets_dev, dets_cpu):
        assert len(ds) == len(cs)
        od, oc = np.lexsort((db[:, 0], dc, -np.round(ds, 3))), np.lexsort((cb[:, 0], cc, -np.round(cs, 3)))
        assert (dc[od] == cc[oc]).mean() > 0.98
        np.testing.assert_allclose(np.sort(ds)[::-1], np.sort(cs)[::-1], atol=2e-4)
