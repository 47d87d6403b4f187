"""CPU tests of the host-side mirror of the reference's plugin surface (no GPU, no kernels):
config, structures, registries, proposal labelling/sampling, box coding, NMS / inference
post-processing (against the oracle's independent numpy restatement) and the image sharding
over ranks (world_size 2, gloo)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import locov_amd
from locov_amd.config import get_cfg
from locov_amd.roi_heads import box_emb_head as beh
from locov_amd.roi_heads import roi_emb_heads as reh
from locov_amd.structures import Boxes, Instances, ShapeSpec, pairwise_iou

LSM_YAML = """
MODEL:
  META_ARCHITECTURE: "DistillProposalMMSSRCNN"
  BACKBONE_PREFIX: ("backbone.body.",)
  LOAD_EMB_PRED_FROM_MMSS_HEAD: True
  RPN:
    POST_NMS_TOPK_TEST: 1000
  ROI_HEADS:
    NAME: "EmbeddingProposalsRes5ROIHeads"
    NUM_CLASSES: 80
    POSITIVE_FRACTION: 1.0
    DETACH_CLASS_PREDICTOR: True
    BATCH_SIZE_PER_IMAGE: 200
  ROI_BOX_HEAD:
    NAME: "EmbeddingFastRCNNOutputLayers"
    CLS_AGNOSTIC_BBOX_REG: True
    EMB_DIM: 768
    EMBEDDING_BASED: True
    FREEZE_EMB_PRED: False
SOLVER:
  IMS_PER_BATCH: 32
"""


@pytest.fixture()
def lsm_cfg(tmp_path):
    p = tmp_path / "coco_lsm.yaml"
    p.write_text(LSM_YAML)
    cfg = get_cfg()
    cfg.merge_from_file(str(p))
    return cfg


def test_config_reads_reference_style_yaml(lsm_cfg):
    cfg = lsm_cfg
    assert cfg.MODEL.ROI_HEADS.NAME == "EmbeddingProposalsRes5ROIHeads"
    assert cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE == 200 and cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION == 1.0
    assert cfg.MODEL.BACKBONE_PREFIX == ("backbone.body.",)             # tuple-in-string is literal-eval'd
    assert cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION == 14 and cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE == "ROIAlignV2"
    assert cfg.SOLVER.IMS_PER_BATCH == 32                                 # unknown sections pass through
    cfg.merge_from_list(["MODEL.ROI_HEADS.NUM_CLASSES", "65", "MODEL.ROI_BOX_HEAD.NORMALIZE_EMB_PRED", "True"])
    assert cfg.MODEL.ROI_HEADS.NUM_CLASSES == 65 and cfg.MODEL.ROI_BOX_HEAD.NORMALIZE_EMB_PRED is True
    c2 = cfg.clone()
    c2.MODEL.ROI_HEADS.NUM_CLASSES = 1
    assert cfg.MODEL.ROI_HEADS.NUM_CLASSES == 65


def test_heads_build_from_config_with_reference_surface(lsm_cfg):
    heads = locov_amd.build_roi_heads(lsm_cfg, {"res4": ShapeSpec(channels=1024, stride=16)})
    assert type(heads).__name__ == "EmbeddingProposalsRes5ROIHeads"
    assert heads.output_shape == 2048 and heads.in_features == ["res4"]
    assert heads.pooler.output_size == (14, 14) and heads.pooler.scales == (1 / 16,) and heads.pooler.aligned
    assert heads.batch_size_per_image == 200 and heads.positive_fraction == 1.0 and heads.proposal_append_gt
    bp = heads.box_predictor
    assert bp.embedding_based and bp.emb_dim == 768 and bp.num_classes is None and bp.cls_score is None
    assert bp.detach_cls_predictor and bp.loss_weight["loss_cls"] == 0.0          # box_emb_head.py:147-149
    assert tuple(bp.emb_pred.weight.shape) == (768, 2048) and tuple(bp.bbox_pred.weight.shape) == (4, 2048)
    keys = set(heads.state_dict().keys())
    for k in ("res5.0.conv1.weight", "res5.0.conv1.norm.running_var", "res5.0.shortcut.weight",
              "res5.2.conv3.norm.bias", "box_predictor.emb_pred.weight", "box_predictor.bbox_pred.bias"):
        assert k in keys, k
    assert not any(k.startswith("res5.1.shortcut") for k in keys)
    # emb_pred parameters can be re-assigned like the meta-arch does (distill_prop_mmss_gcnn.py:121-125)
    v2l = torch.nn.Linear(2048, 768)
    bp.emb_pred.weight, bp.emb_pred.bias = v2l.weight, v2l.bias
    assert bp.emb_pred.weight is v2l.weight
    # bank install: trainer.py:365-396 contract (CPU tensors here; normalisation off -> no kernel call)
    bank = np.zeros((66, 768), np.float32)
    bank[:65] = np.random.default_rng(0).standard_normal((65, 768)) * 0.05
    bp.set_class_embeddings(bank)
    assert bp.num_classes == 65 and tuple(bp.cls_score.weight.shape) == (66, 768)
    assert not bp.cls_score.weight.requires_grad and torch.all(bp.cls_score.bias == 0)
    assert "box_predictor.cls_score.weight" in heads.state_dict()
    # the multi-token predictor (box_emb_grounding_head.py:259) is built through the same name lookup
    lsm_cfg.MODEL.ROI_BOX_HEAD.NAME = "EmbeddingGroundingFastRCNNOutputLayers"
    gp = beh.build_box_predictor(lsm_cfg, 2048)
    assert type(gp).__name__ == "EmbeddingGroundingFastRCNNOutputLayers" and gp.num_classes is None
    assert type(gp.cls_score).__name__ == "GroundingModule" and gp.cls_score.temperature == 10.0
    assert tuple(gp.emb_pred.weight.shape) == (768, 2048) and gp.loss_weight["loss_cls"] == 0.0
    with pytest.raises(KeyError):
        lsm_cfg.MODEL.ROI_BOX_HEAD.NAME = "NoSuchOutputLayers"
        beh.build_box_predictor(lsm_cfg, 2048)


def test_forward_before_bank_is_an_error(lsm_cfg):
    bp = beh.build_box_predictor(lsm_cfg, 2048)
    with pytest.raises(RuntimeError, match="set_class_embeddings"):
        bp(torch.zeros(2, 2048))


def test_structures():
    b = Boxes(torch.tensor([[0., 0., 10., 10.], [5., 5., 15., 25.], [-5., -5., 3., 3.]]))
    assert torch.equal(b.area(), torch.tensor([100., 200., 64.]))
    np.testing.assert_allclose(pairwise_iou(b, b).diag().numpy(), 1.0)
    assert pairwise_iou(b[:1], b[1:2]).item() == pytest.approx(25.0 / 275.0)
    b.clip((20, 12))
    assert b.tensor[1].tolist() == [5., 5., 12., 20.] and b.tensor[2].tolist() == [0., 0., 3., 3.]
    assert len(Boxes(torch.zeros(0, 4))) == 0 and len(Boxes.cat([b, b])) == 6
    inst = Instances((20, 12), proposal_boxes=b, objectness_logits=torch.arange(3.))
    assert len(inst) == 3 and inst.has("proposal_boxes") and inst.image_size == (20, 12)
    sub = inst[torch.tensor([2, 0])]
    assert sub.objectness_logits.tolist() == [2., 0.] and len(sub.proposal_boxes) == 2
    with pytest.raises(AssertionError):
        inst.set("bad", torch.zeros(5))
    cat = Instances.cat([inst, sub])
    assert len(cat) == 5 and isinstance(cat.proposal_boxes, Boxes)


def test_matcher_and_sampling():
    m = reh.Matcher([0.5], [0, 1])
    iou = torch.tensor([[0.1, 0.6, 0.5, 0.0], [0.7, 0.2, 0.49, 0.0]])
    idx, lab = m(iou)
    assert idx.tolist() == [1, 0, 0, 0] and lab.tolist() == [1, 1, 1, 0]
    idx, lab = m(torch.zeros(0, 3))
    assert idx.tolist() == [0, 0, 0] and lab.tolist() == [0, 0, 0]
    torch.manual_seed(0)
    labels = torch.tensor([3, 80, 80, 1, -1, 80, 7, 80])
    pos, neg = reh.subsample_labels(labels, 4, 0.5, 80)
    assert len(pos) == 2 and len(neg) == 2 and set(pos.tolist()) <= {0, 3, 6} and set(neg.tolist()) <= {1, 2, 5, 7}
    pos, neg = reh.subsample_labels(labels, 100, 1.0, 80)
    assert len(pos) == 3 and len(neg) == 4


def test_subsample_order_is_the_sync_free_form_of_subsample_labels():
    """subsample_order: the candidates in random order with the foreground (resp. background) ones first + the population
    sizes; its first num_pos / num_neg entries are what subsample_labels would draw (same sets, uniform subsets)."""
    labels = torch.tensor([3, 80, 80, 1, -1, 80, 7, 80, -1, 80])
    torch.manual_seed(1)
    pos_order, neg_order, counts = reh.subsample_order(labels, 80)
    assert counts.tolist() == [3, 5]
    assert sorted(pos_order[:3].tolist()) == [0, 3, 6] and sorted(neg_order[:5].tolist()) == [1, 2, 5, 7, 9]
    assert sorted(pos_order.tolist()) == list(range(10)) and sorted(neg_order.tolist()) == list(range(10))
    # every background candidate is drawn first about equally often
    hits = torch.zeros(10)
    for _ in range(400):
        hits[reh.subsample_order(labels, 80)[1][0]] += 1
    assert hits[[0, 3, 4, 6, 8]].sum() == 0 and hits[[1, 2, 5, 7, 9]].min() > 40


def test_box_reg_loss_by_mask_equals_the_indexed_form(lsm_cfg):
    """FastRCNNOutputLayers.box_reg_loss selects the foreground rows by a mask (no nonzero, no host sync): same value and
    gradient as Detectron2's indexed form, also with degenerate background boxes."""
    from locov_amd.roi_heads import box_emb_head as beh
    pred = locov_amd.build_box_predictor(lsm_cfg, ShapeSpec(channels=64)) if hasattr(locov_amd, "build_box_predictor") else \
        beh.build_box_predictor(lsm_cfg, ShapeSpec(channels=64))
    pred.num_classes = 80
    g = torch.Generator().manual_seed(3)
    R = 50
    xy = torch.rand(R, 2, generator=g) * 500
    boxes = torch.cat([xy, xy + 20 + torch.rand(R, 2, generator=g) * 100], 1)
    boxes[7] = torch.tensor([10.0, 10.0, 10.0, 10.0])                 # a degenerate BACKGROUND proposal
    gt = boxes + torch.randn(R, 4, generator=g) * 5
    classes = torch.randint(0, 81, (R,), generator=g)
    classes[7] = 80
    classes[:5] = torch.tensor([0, 5, 79, 80, 80])
    deltas = torch.randn(R, 4, generator=g, requires_grad=True)
    loss = pred.box_reg_loss(boxes, gt, deltas, classes)
    loss.backward()
    fg = torch.nonzero((classes >= 0) & (classes < 80), as_tuple=True)[0]
    d2 = deltas.detach().clone().requires_grad_(True)
    want = beh.smooth_l1_loss(d2[fg], pred.box2box_transform.get_deltas(boxes[fg], gt[fg]), pred.smooth_l1_beta, reduction="sum") / R
    want.backward()
    assert abs(float(loss) - float(want)) <= 1e-6 * max(1.0, abs(float(want)))
    assert torch.allclose(deltas.grad, d2.grad, atol=1e-7) and float(deltas.grad[7].abs().max()) == 0.0


def test_label_and_sample_proposals_contract(lsm_cfg):
    lsm_cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 16
    heads = locov_amd.build_roi_heads(lsm_cfg, {"res4": ShapeSpec(channels=1024, stride=16)})
    heads.num_classes = 80
    rng = np.random.default_rng(0)
    props, tgts = [], []
    for n_gt in (3, 0):
        p = Instances((800, 1333))
        xy = rng.uniform(0, 600, (40, 2)).astype(np.float32)
        p.proposal_boxes = Boxes(torch.from_numpy(np.concatenate([xy, xy + rng.uniform(20, 200, (40, 2)).astype(np.float32)], 1)))
        p.objectness_logits = torch.zeros(40)
        t = Instances((800, 1333))
        t.gt_boxes = Boxes(p.proposal_boxes.tensor[:n_gt] + 1.0)
        t.gt_classes = torch.ones(n_gt, dtype=torch.int64)
        props.append(p)
        tgts.append(t)
    out = heads.label_and_sample_proposals(props, tgts)
    assert len(out) == 2
    assert len(out[0]) <= 16 and out[0].has("gt_boxes") and out[0].has("fg_proposal")
    fg = out[0].gt_classes != 80
    assert fg.sum() >= 3                                          # GT boxes are appended to the proposals
    assert torch.equal(out[0].fg_proposal, fg.to(out[0].fg_proposal.dtype))
    assert torch.all(out[1].gt_classes == 80) and not out[1].has("gt_boxes")     # image without GT
    ev = reh.get_event_storage()
    assert "roi_head/num_fg_samples" in ev.scalars and "roi_head/num_bg_samples" in ev.scalars


def test_box2box_transform_round_trip_and_oracle(oracle):
    t = beh.Box2BoxTransform((10.0, 10.0, 5.0, 5.0))
    rng = np.random.default_rng(1)
    src = oracle.synth_boxes(rng, 50)
    dst = oracle.synth_boxes(rng, 50)
    keep = ((src[:, 2] - src[:, 0]) > 1) & ((src[:, 3] - src[:, 1]) > 1) & ((dst[:, 2] - dst[:, 0]) > 1) & ((dst[:, 3] - dst[:, 1]) > 1)
    src, dst = torch.from_numpy(src[keep]), torch.from_numpy(dst[keep])
    deltas = t.get_deltas(src, dst)
    np.testing.assert_allclose(t.apply_deltas(deltas, src).numpy(), dst.numpy(), rtol=1e-4, atol=1e-2)
    d = (rng.standard_normal((len(src), 4)) * 3).astype(np.float32)
    d[0, 2:] = 50.0                                                # hits the log(1000/16) clamp
    np.testing.assert_allclose(t.apply_deltas(torch.from_numpy(d), src).numpy(), oracle.apply_deltas(d, src.numpy()),
                               rtol=1e-5, atol=1e-3)


def test_nms_and_inference_match_oracle(oracle):
    rng = np.random.default_rng(2)
    boxes = oracle.synth_boxes(rng, 300)
    scores = rng.uniform(0, 1, 300).astype(np.float32)
    keep = beh.nms(torch.from_numpy(boxes), torch.from_numpy(scores), 0.5).numpy()
    np.testing.assert_array_equal(keep, oracle.nms(boxes, scores, 0.5))
    cls = rng.integers(0, 5, 300)
    keep = beh.batched_nms(torch.from_numpy(boxes), torch.from_numpy(scores), torch.from_numpy(cls), 0.5).numpy()
    np.testing.assert_array_equal(keep, oracle.batched_nms(boxes, scores, cls, 0.5))
    assert beh.batched_nms(torch.zeros(0, 4), torch.zeros(0), torch.zeros(0, dtype=torch.int64), 0.5).numel() == 0
    probs = oracle.softmax(rng.standard_normal((300, 9)).astype(np.float32) * 3)
    probs[5, 0] = np.nan                                           # non-finite rows are dropped
    inst, kept = beh.fast_rcnn_inference_single_image(torch.from_numpy(boxes), torch.from_numpy(probs), (800, 1333),
                                                      0.05, 0.5, 100)
    wb, ws, wc = oracle.fast_rcnn_inference_single_image(boxes, probs, (800, 1333), 0.05, 0.5, 100)
    np.testing.assert_allclose(inst.pred_boxes.tensor.numpy(), wb, atol=1e-4)
    np.testing.assert_allclose(inst.scores.numpy(), ws, atol=1e-6)
    np.testing.assert_array_equal(inst.pred_classes.numpy(), wc)
    assert len(inst) <= 100


def test_losses_follow_detectron2_definition(lsm_cfg):
    lsm_cfg.MODEL.ROI_HEADS.DETACH_CLASS_PREDICTOR = False
    bp = beh.build_box_predictor(lsm_cfg, 64)
    bp.num_classes = 80
    g = torch.Generator().manual_seed(0)
    scores = torch.randn(6, 81, generator=g, requires_grad=True)
    deltas = torch.randn(6, 4, generator=g, requires_grad=True)
    p = Instances((100, 100))
    p.proposal_boxes = Boxes(torch.tensor([[10., 10., 50., 50.]] * 6))
    p.gt_boxes = Boxes(torch.tensor([[12., 8., 48., 55.]] * 6))
    p.gt_classes = torch.tensor([1, 80, 80, 3, 80, 80])
    losses = bp.losses((scores, deltas), [p])
    want_cls = torch.nn.functional.cross_entropy(scores, p.gt_classes)
    tgt = bp.box2box_transform.get_deltas(p.proposal_boxes.tensor[[0, 3]], p.gt_boxes.tensor[[0, 3]])
    want_box = (deltas[[0, 3]] - tgt).abs().sum() / 6.0             # smooth-L1 beta=0, sum / #proposals
    assert torch.allclose(losses["loss_cls"], want_cls) and torch.allclose(losses["loss_box_reg"], want_box)
    empty = bp.losses((scores[:0], deltas[:0]), [])
    assert float(empty["loss_cls"]) == 0.0 and float(empty["loss_box_reg"]) == 0.0


def test_registry_and_configurable():
    assert "EmbeddingRes5ROIHeads" in reh.ROI_HEADS_REGISTRY and "EmbeddingProposalsRes5ROIHeads" in reh.ROI_HEADS_REGISTRY
    with pytest.raises(KeyError):
        reh.ROI_HEADS_REGISTRY.get("StandardROIHeads")
    from locov_amd.roi_heads import register_with_detectron2
    assert register_with_detectron2() is False                     # Detectron2 is not installed here
    from locov_amd.poolers import ROIPooler
    with pytest.raises(ValueError):
        ROIPooler(7, (1 / 16,), 0, "ROIPool")
    p = ROIPooler(7, (1 / 4, 1 / 8, 1 / 16, 1 / 32), 0, "ROIAlignV2")
    assert (p.min_level, p.max_level) == (2, 5)
    with pytest.raises(AssertionError):
        ROIPooler(7, (1 / 4, 1 / 16), 0, "ROIAlignV2")             # not a pyramid


# ------------------------------------------------------------------ image sharding, world_size 2 (gloo)
def test_shard_range_tiles_the_image_list():
    from locov_amd.sharding import shard_range
    for n in (0, 1, 7, 8, 9, 31):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_images, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from locov_amd.sharding import gather_per_image, max_over_ranks, shard_list, world_info
    assert world_info() == (rank, world)
    images = list(range(n_images))
    mine = shard_list(images, rank, world)
    # stand-in for the per-image head output: (image id, number of proposals scored)
    local = [(i, 100 + i) for i in mine]
    allres = gather_per_image(local, n_images)
    assert [r[0] for r in allres] == images and [r[1] for r in allres] == [100 + i for i in images]
    t = max_over_ranks(1.0 + rank)                      # slowest rank defines the job time
    assert t == float(world)
    from locov_amd.sharding import all_ranks
    assert all_ranks(10.0 + rank) == [10.0 + r for r in range(world)]      # the bench line's per_rank_ms_per_step, rank order
    total = torch.tensor([sum(r[1] for r in local)], dtype=torch.float64)
    dist.all_reduce(total)                              # whole-job proposal count = sum over ranks
    assert total.item() == sum(100 + i for i in images)
    dist.barrier()
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


def test_image_sharding_world_size_2_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, 7, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_bench_gpus_n_without_launcher_starts_n_ranks_or_fails_loudly():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start two ranks through torch.distributed.run as a child
    process (never run one rank and report n_gpus 1).  Without a GPU the ranks stop with the no-fallback message and the
    parent relays a non-zero exit code."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    if torch.cuda.is_available():
        return                       # on the GPU box the multi-rank run itself is covered by tests/test_gpu_multirank.py
    assert r.returncode != 0
    assert "2-rank child exited" in r.stderr
    assert r.stderr.count("bench.py needs a ROCm GPU") >= 1 or "needs a ROCm GPU" in r.stdout + r.stderr


def test_batched_nms_per_class_fallback_equals_the_single_launch(monkeypatch):
    """Above 40 000 candidates batched_nms runs per class (Detectron2's rule; the K x K/64 bit matrix would not fit);
    both forms must keep the same boxes in the same order."""
    from locov_amd.roi_heads import box_emb_head as beh
    g = torch.Generator().manual_seed(11)
    n = 600
    xy = torch.rand(n, 2, generator=g) * 200
    wh = torch.rand(n, 2, generator=g) * 60 + 4
    boxes = torch.cat([xy, xy + wh], dim=1)
    scores = torch.rand(n, generator=g)
    idxs = torch.randint(0, 7, (n,), generator=g)
    one = beh.batched_nms(boxes, scores, idxs, 0.5)
    monkeypatch.setattr(beh, "_PER_CLASS_NMS_ABOVE", 100)
    per_class = beh.batched_nms(boxes, scores, idxs, 0.5)
    assert torch.equal(one, per_class)
    assert beh._PER_CLASS_NMS_ABOVE == 100


def test_bench_refuses_a_pmc_traffic_record_taken_on_other_sources(monkeypatch):
    """bench.py's `roofline.traffic` comes from a committed PMC pass (counters cannot be collected inside the timed run).  The record
    names the kernel sources it was taken on; for the committed tree it must be current, and for a library built from any other
    sources -- or another workload -- it reads as None, never as a stale number."""
    import importlib, json, sys, types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    from locov_amd import build
    path = os.path.join(root, "profiles", bench.TRAFFIC_FILE)
    if not os.path.exists(path):                      # (no PMC pass committed for this round yet: said so, loudly, in the line)
        args = bench.parse([])
        assert bench.traffic_record(args)[0] is None and "unreadable" in bench.traffic_source(args)
        assert bench.recorded_traffic(args, "gemm_split") is None
        return
    rec = json.load(open(path))
    wl = rec["workload"]
    args = types.SimpleNamespace(**{k: wl[k] for k in ("images", "proposals", "classes", "dim", "res5", "conv3x3", "block0", "res5_dtype")})
    if rec.get("source_fingerprint") == build.source_fingerprint():       # (a tree whose kernels changed after the last PMC pass: already None)
        got = bench.recorded_traffic(args, "gemm_split")
        assert got is not None and 1e9 < got < 2e10
        assert "ANOTHER box" in bench.traffic_source(args)
    else:
        assert "STALE" in bench.traffic_source(args) and bench.recorded_traffic(args, "gemm_split") is None
    monkeypatch.setattr(build, "source_fingerprint", lambda: "some other tree")
    assert bench.recorded_traffic(args, "gemm_split") is None
    assert bench.traffic_source(args).startswith("none -- ") and "STALE" in bench.traffic_source(args)
    monkeypatch.undo()
    args.proposals = wl["proposals"] + 1
    assert bench.recorded_traffic(args, "gemm_split") is None
    assert "another workload (differs in proposals)" in bench.traffic_source(args)


def test_stock_library_fallback_is_announced_once_with_the_failed_condition(lsm_cfg):
    """with RES5_BACKEND "hip" a call that cannot take the hand-written Res5 path (odd pooler size,
    non-FrozenBN, grouped 3x3, channels not a multiple of 32, host tensors, a dtype without a backward) runs torch.conv2d --
    that must be said once, naming the condition; "miopen" as the configured backend is a choice and stays silent."""
    import warnings
    heads = locov_amd.build_roi_heads(lsm_cfg, {"res4": ShapeSpec(channels=1024, stride=16)})
    feat = torch.zeros(1, 1024, 8, 8)
    assert heads._rows_path_reason([feat]) == "the feature map is on cpu"
    with pytest.warns(RuntimeWarning, match=r"_shared_roi_transform: RES5_BACKEND is 'hip' but .*the feature map is on cpu"):
        heads._warn_stock_fallback("_shared_roi_transform", [feat])
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                   # the second time: nothing
        heads._warn_stock_fallback("_shared_roi_transform", [feat])
    with pytest.warns(RuntimeWarning, match="_res5_grid"):               # another call site is its own announcement
        heads._warn_stock_fallback("_res5_grid", [feat])
    assert "1000 input channels" in heads._rows_path_reason([torch.zeros(1, 1000, 8, 8)])
    assert "2 feature levels" in heads._rows_path_reason([feat, feat])
    heads.pooler.output_size = (7, 7)
    assert "7x7 is not square and even" in heads._rows_path_reason([feat])
    heads.pooler.output_size = (14, 14)
    heads.res5[1].conv2.groups = 2
    assert "ungrouped" in heads._rows_path_reason([feat])
    heads.res5[1].conv2.groups = 1
    heads.res5_backend = "miopen"
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        heads._warn_stock_fallback("forward", [feat])
    # and the call site really warns before it takes the stock path (the pooler then refuses host tensors: no CPU fallback below it)
    heads.res5_backend = "hip"
    heads.__dict__.pop("_fallback_warned", None)
    with pytest.warns(RuntimeWarning, match="stock library path"):
        with pytest.raises(Exception):
            heads._shared_roi_transform([feat], [Boxes(torch.tensor([[0.0, 0.0, 32.0, 32.0]]))])


def test_cat_rows_is_a_view_of_adjacent_pieces_and_a_copy_otherwise():
    """structures.cat_rows: the concatenation of per-image views of one batch-wide tensor costs no launch; anything else is torch.cat."""
    import torch
    from locov_amd.structures import cat_rows
    b = torch.arange(40.).view(10, 4)
    v = cat_rows([b[2:5], b[5:9]])
    assert v.data_ptr() == b[2:].data_ptr() and torch.equal(v, b[2:9])
    v = cat_rows([b[2:5], b[6:9]])                                   # a gap: a copy
    assert torch.equal(v, torch.cat([b[2:5], b[6:9]])) and v.data_ptr() != b[2:].data_ptr()
    parts = torch.split(torch.arange(12), [3, 4, 5])
    v = cat_rows(list(parts))
    assert torch.equal(v, torch.arange(12)) and v.data_ptr() == parts[0].data_ptr()
    assert cat_rows([torch.ones(3), torch.zeros(2)]).tolist() == [1, 1, 1, 0, 0]             # two storages
    assert torch.equal(cat_rows([b[:, :2][1:3], b[:, :2][3:4]]), b[1:4, :2])                  # non-contiguous pieces
    assert torch.equal(cat_rows([b[0:2], b[2:2], b[2:4]]), b[0:4])                            # an empty piece in between
    c = torch.arange(24.).view(6, 4)
    assert torch.equal(cat_rows([b[0:2], c[2:3]]), torch.cat([b[0:2], c[2:3]]))
    g = torch.ones(4, 2, requires_grad=True) * 1.0
    assert cat_rows([g[0:2], g[2:4]]).requires_grad                                           # autograd tensors: torch.cat
    assert cat_rows([b[1:3]]).data_ptr() == b[1:3].data_ptr()



def test_bench_training_batches_of_the_references_real_shapes():
    """bench.synth_train_batch / synth_image_size: image sizes out of configs/coco_stt.yaml:54's MIN_SIZE_TRAIN with
    the long side capped at 1333, the batch's res4 map = ceil(largest image / 16) per axis, per-image proposal / ground-truth counts,
    an image that cannot fill the sampling budget, boxes inside their own image."""
    import importlib, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    gen = torch.Generator().manual_seed(3)
    sizes = [bench.synth_image_size(gen) for _ in range(500)]
    assert all(min(h, w) <= 800 and max(h, w) <= 1333 for h, w in sizes)
    assert {min(h, w) for h, w in sizes if max(h, w) < 1333} <= set(bench.MIN_SIZE_TRAIN)
    assert any(h > w for h, w in sizes) and any(w > h for h, w in sizes)                         # portrait and landscape
    b = bench.synth_train_batch(gen, "cpu", 4, 300, 80, 32, multiscale=True, max_gt=15, fixed_gt=False, short_image=(2, 17))
    hm, wm = max(h for h, _ in b["sizes"]), max(w for _, w in b["sizes"])
    assert tuple(b["features"].shape) == (4, 1024, -(-hm // 16), -(-wm // 16)) == (4, 1024) + b["map"]
    assert [len(p) for p in b["proposals"]] == [300, 300, 17, 300]
    assert all(0 <= len(t) <= 15 for t in b["targets"])
    for p, t, (h, w) in zip(b["proposals"], b["targets"], b["sizes"]):
        assert p.image_size == (h, w) == t.image_size
        box = p.proposal_boxes.tensor
        assert float(box[:, 0].min()) >= 0 and float(box[:, 2].max()) <= w + 1 and float(box[:, 3].max()) <= h + 1
        assert bool(((box[:, 2] - box[:, 0]) > 0).all() and ((box[:, 3] - box[:, 1]) > 0).all())
    fixed = bench.synth_train_batch(torch.Generator().manual_seed(5), "cpu", 2, 50, 80, 32)
    assert tuple(fixed["features"].shape) == (2, 1024, 50, 84) and [len(t) for t in fixed["targets"]] == [7, 7]
    crowded = bench.synth_train_batch(torch.Generator().manual_seed(6), "cpu", 2, 64, 80, 32, multiscale=True, crowded_image=1)
    from locov_amd.structures import pairwise_iou
    iou = pairwise_iou(crowded["targets"][1].gt_boxes, crowded["proposals"][1].proposal_boxes)
    assert float(iou.max(dim=0).values.min()) > 0.5                                                 # every proposal sits on a GT box
