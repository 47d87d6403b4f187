"""Row a-11 against the oracle: SampleAllROIHeads.label_and_sample_proposals
(ovr/modeling/roi_heads/roi_emb_heads.py:25-118) and box_predictor.losses (:266,:347) vs the numpy restatement of
Detectron2's definitions in oracle/lsm_oracle.py -- on CPU tensors here and, with the same code, on the device
(`-m gpu`): which proposals may be drawn and how many, every sampled proposal's class / foreground flag / matched
ground-truth fields, and both losses."""
import numpy as np
import pytest
import torch

from locov_amd.config import get_cfg
from locov_amd.structures import Boxes, Instances, ShapeSpec


def _heads(detach, batch, pos_frac, device):
    import locov_amd
    cfg = get_cfg()
    cfg.MODEL.RESNETS.RES2_OUT_CHANNELS = 32
    cfg.MODEL.RESNETS.WIDTH_PER_GROUP = 8
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    cfg.MODEL.ROI_BOX_HEAD.EMB_DIM = 96
    cfg.MODEL.ROI_HEADS.NAME = "EmbeddingProposalsRes5ROIHeads"
    cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = batch
    cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION = pos_frac
    cfg.MODEL.ROI_HEADS.DETACH_CLASS_PREDICTOR = detach
    torch.manual_seed(0)
    heads = locov_amd.build_roi_heads(cfg, {"res4": ShapeSpec(channels=128, stride=16)}).to(device).train()
    bank = np.zeros((81, 96), np.float32)
    bank[:80] = np.random.default_rng(0).standard_normal((80, 96)) * 0.05
    heads.box_predictor.set_class_embeddings(bank)
    heads.num_classes = 80
    return heads


def _batch(oracle, rng, device, n_img=3, r=120, n_gt=6):
    props, targets, raw = [], [], []
    for i in range(n_img):
        g = n_gt if i != 1 else 0                                   # one image without ground truth
        gt = oracle.synth_boxes(rng, g) if g else np.zeros((0, 4), np.float32)
        b = oracle.synth_boxes(rng, r)
        if g:
            jitter = rng.uniform(-12, 12, (3 * g, 4)).astype(np.float32)
            b[:3 * g] = np.repeat(gt, 3, axis=0) + jitter            # candidates around the ground truth: IoU on both sides of 0.5
            b[:, 2:] = np.maximum(b[:, 2:], b[:, :2] + 1.0)
        cls = rng.integers(0, 80, g)
        p = Instances((800, 1333))
        p.proposal_boxes = Boxes(torch.from_numpy(b).to(device))
        p.objectness_logits = torch.from_numpy(rng.standard_normal(r).astype(np.float32)).to(device)
        t = Instances((800, 1333))
        t.gt_boxes = Boxes(torch.from_numpy(gt).to(device))
        t.gt_classes = torch.from_numpy(cls).to(device)
        props.append(p)
        targets.append(t)
        raw.append((b, gt, cls))
    return props, targets, raw


def _check(oracle, device, detach, batch, pos_frac):
    heads = _heads(detach, batch, pos_frac, device)
    rng = np.random.default_rng(batch)
    props, targets, raw = _batch(oracle, rng, device)
    torch.manual_seed(5)
    sampled = heads.label_and_sample_proposals(props, targets)
    assert len(sampled) == len(props)
    all_boxes, all_gt, all_cls = [], [], []
    for s, (b, gt, cls) in zip(sampled, raw):
        cand, idx, labels, best = oracle.label_proposals(b, gt, cls, 80)
        n_fg, n_bg = oracle.expected_sample_counts(labels, 80, batch, pos_frac)
        got_boxes = s.proposal_boxes.tensor.cpu().numpy()
        got_cls = s.gt_classes.cpu().numpy()
        assert len(s) == n_fg + n_bg and int((got_cls != 80).sum()) == n_fg and int((got_cls == 80).sum()) == n_bg
        np.testing.assert_array_equal(s.fg_proposal.cpu().numpy(), (got_cls != 80).astype(np.int64))
        # every sampled row is one of the image's candidates, drawn at most once, with the oracle's label and matched gt
        key = {tuple(np.round(c, 4)): j for j, c in enumerate(cand)}       # (duplicates among candidates do not occur here)
        assert len(key) == len(cand)
        rows = [key[tuple(np.round(x, 4))] for x in got_boxes]
        assert len(set(rows)) == len(rows)
        np.testing.assert_array_equal(got_cls, labels[rows])
        if len(gt):
            np.testing.assert_array_equal(s.gt_boxes.tensor.cpu().numpy(), gt[idx[rows]])       # ALL target fields are copied (:97-100)
            assert s.has("gt_classes") and s.has("objectness_logits")
        # foreground rows have IoU >= 0.5 with their ground truth, background rows < 0.5
        assert np.all(best[rows][got_cls != 80] >= 0.5) and np.all(best[rows][got_cls == 80] < 0.5)
        all_boxes.append(got_boxes)
        all_gt.append(s.gt_boxes.tensor.cpu().numpy() if s.has("gt_boxes") else got_boxes)
        all_cls.append(got_cls)
    # losses of the predictor on those proposals (predictions are what they are: random)
    n = sum(len(s) for s in sampled)
    g = torch.Generator().manual_seed(9)
    scores = (torch.randn(n, 81, generator=g) * 2).to(device).requires_grad_(True)
    deltas = (torch.randn(n, 4, generator=g) * 0.3).to(device).requires_grad_(True)
    got = heads.box_predictor.losses((scores, deltas), sampled)
    want = oracle.box_head_losses(scores.detach().cpu().numpy(), deltas.detach().cpu().numpy(), np.concatenate(all_boxes),
                                  np.concatenate(all_gt), np.concatenate(all_cls), 80, 0.0 if detach else 1.0)
    assert set(got) == {"loss_cls", "loss_box_reg"}
    assert abs(float(got["loss_cls"]) - want["loss_cls"]) <= 1e-5 * max(1.0, abs(want["loss_cls"]))
    assert abs(float(got["loss_box_reg"]) - want["loss_box_reg"]) <= 1e-5 * max(1.0, abs(want["loss_box_reg"]))


@pytest.mark.parametrize("detach,batch,pos_frac", [(True, 64, 1.0), (False, 200, 0.25), (False, 16, 0.5)])
def test_label_sample_and_losses_vs_oracle_cpu(oracle, detach, batch, pos_frac):
    _check(oracle, "cpu", detach, batch, pos_frac)


@pytest.mark.gpu
@pytest.mark.parametrize("detach,batch,pos_frac", [(True, 64, 1.0), (False, 200, 0.25)])
def test_label_sample_and_losses_vs_oracle_gpu(oracle, detach, batch, pos_frac):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device")
    _check(oracle, "cuda", detach, batch, pos_frac)


def test_labelling_raises_the_references_asserts_on_the_host(oracle):
    """Detectron2's two host-side validity asserts on this path -- Matcher: `torch.all(match_quality_matrix >= 0)`,
    Box2BoxTransform.get_deltas: `(src_widths > 0).all()` "Input boxes to Box2BoxTransform are not valid!" -- are raised as
    AssertionError by label_and_sample_proposals' ONE host read (device-side asserts are compiled out of this ROCm build):
    a NaN proposal box has NaN IoU (fails `>= 0`), and a labelled-foreground box that is not strictly positive in width and
    height fails the box check."""
    heads = _heads(True, 64, 1.0, "cpu")
    rng = np.random.default_rng(3)
    props, targets, _ = _batch(oracle, rng, "cpu")
    heads.label_and_sample_proposals(props, targets)                       # valid input: no error
    # (a NaN proposal box does NOT trip either assert, here or upstream: pairwise_iou is 0 where the intersection is not
    # positive, so such a box is background and never reaches get_deltas)
    bad = props[0].proposal_boxes.tensor.clone()
    bad[5, 0] = float("nan")
    props[0].proposal_boxes = Boxes(bad)
    heads.label_and_sample_proposals(props, targets)
    # Matcher's assert: a match-quality matrix with a negative (or NaN) entry
    import locov_amd.roi_heads.labelling as reh
    real_iou = reh.pairwise_iou
    for poison in (-0.25, float("nan")):
        def fake(a, b, poison=poison):
            q = real_iou(a, b).clone()
            if q.numel():
                q.view(-1)[3] = poison
            return q
        reh.pairwise_iou = fake
        try:
            with pytest.raises(AssertionError, match="match quality"):
                heads.label_and_sample_proposals(props, targets)
        finally:
            reh.pairwise_iou = real_iou
    # the box check on its own: labels say foreground, the box has zero width (fed past the matcher)
    props, targets, _ = _batch(oracle, rng, "cpu")
    m = heads._match_one_image(props[0], targets[0])
    assert m[-1].shape == (4,) and int(m[-1][2]) == 0 and int(m[-1][3]) == 0
    gt = targets[0].gt_boxes.tensor
    squashed = props[0].proposal_boxes.tensor.clone()
    squashed[0] = gt[0]
    squashed[0, 2] = squashed[0, 0]                                        # zero width ...
    props[0].proposal_boxes = Boxes(squashed)
    heads.proposal_matcher_orig = heads.proposal_matcher
    class _AllFg:                                                          # ... and a matcher that still calls row 0 foreground
        check_quality = True
        def __call__(self, q):
            idx, lab = heads.proposal_matcher_orig(q)
            lab = lab.clone(); lab[0] = 1
            return idx, lab
    heads.proposal_matcher = _AllFg()
    with pytest.raises(AssertionError, match="Input boxes to Box2BoxTransform are not valid"):
        heads.label_and_sample_proposals(props, targets)
    # get_deltas keeps the reference's own assert for every other caller
    from locov_amd.roi_heads.box_emb_head import Box2BoxTransform
    t = Box2BoxTransform((10.0, 10.0, 5.0, 5.0))
    with pytest.raises(AssertionError, match="not valid"):
        t.get_deltas(torch.tensor([[0.0, 0.0, 0.0, 5.0]]), torch.tensor([[0.0, 0.0, 4.0, 5.0]]))


def test_box_reg_loss_ignores_non_finite_background_predictions():
    """The mask-based box regression loss never lets a background / ignored row reach the value or the gradient: an inf or
    NaN prediction there (which the indexed upstream form does not touch) leaves both exactly as without it."""
    heads = _heads(False, 16, 0.5, "cpu")
    bp = heads.box_predictor
    g = torch.Generator().manual_seed(1)
    n = 12
    boxes = torch.rand(n, 4, generator=g) * 100
    boxes[:, 2:] += boxes[:, :2] + 5
    gt = boxes + torch.randn(n, 4, generator=g)
    cls = torch.tensor([3, 80, 80, 7, -1, 80, 1, 80, 80, -1, 5, 80])
    pred = (torch.randn(n, 4, generator=g) * 0.2)
    clean = pred.clone().requires_grad_(True)
    want = bp.box_reg_loss(boxes, gt, clean, cls)
    want.backward()
    dirty = pred.clone()
    dirty[1] = float("inf"); dirty[4] = float("nan"); dirty[8, 2] = -float("inf")
    dirty.requires_grad_(True)
    got = bp.box_reg_loss(boxes, gt, dirty, cls)
    got.backward()
    assert torch.isfinite(got) and float(got) == float(want)
    assert torch.isfinite(dirty.grad).all() and torch.equal(dirty.grad, clean.grad)
    assert not dirty.grad[[1, 2, 4, 5, 7, 8, 9, 11]].any()


def test_batched_labelling_equals_the_per_image_form(oracle):
    """label_and_sample_proposals labels the whole batch in one set of launches (IoU of all targets x all proposals, pairs of
    different images masked out).  With a budget that takes EVERY candidate the draw is no longer random, so the batch form and
    the per-image form must return the same sets: rows, classes, foreground flags and every copied target field."""
    heads = _heads(True, 10 ** 6, 1.0, "cpu")
    rng = np.random.default_rng(11)
    props, targets, _ = _batch(oracle, rng, "cpu", n_img=4, r=90, n_gt=5)
    for t in targets:                                            # an extra target field must travel too (:97-100)
        t.set("gt_tag", torch.arange(len(t), dtype=torch.float32) + 0.5)
    a = heads.label_and_sample_proposals(props, targets)
    from locov_amd.roi_heads.roi_emb_heads import add_ground_truth_to_proposals
    b = heads._label_and_sample_per_image(add_ground_truth_to_proposals(targets, props), targets)

    def canon(inst):
        order = np.lexsort(inst.proposal_boxes.tensor.numpy().T[::-1])
        return {k: (v.tensor if hasattr(v, "tensor") else v).numpy()[order] for k, v in inst.get_fields().items()}
    for x, y, t in zip(a, b, targets):
        cx, cy = canon(x), canon(y)
        assert set(cx) == set(cy) and len(x) == len(y)
        for k in cx:
            if len(t) == 0 and k in ("gt_boxes", "gt_tag"):
                continue
            np.testing.assert_array_equal(cx[k], cy[k], err_msg=k)
    assert not a[1].has("gt_boxes") and not b[1].has("gt_boxes")          # the image without ground truth carries no target fields


def _three_interval_check(oracle, device):
    """IOU_THRESHOLDS [0.3, 0.7] / IOU_LABELS [-1, 0, 1] (the Matcher's "no match" label is IGNORE): an image without ground truth
    inside a batch that has some still has every proposal labelled BACKGROUND ([D2-upstream] ROIHeads._sample_proposals,
    has_gt == False), not ignored -- the batch form against the per-image form, which is the reference's code path."""
    from locov_amd.roi_heads.roi_emb_heads import Matcher, add_ground_truth_to_proposals
    heads = _heads(True, 10 ** 6, 1.0, device)
    heads.proposal_matcher = Matcher([0.3, 0.7], [-1, 0, 1], allow_low_quality_matches=False)
    rng = np.random.default_rng(17)
    props, targets, _ = _batch(oracle, rng, device, n_img=3, r=70, n_gt=4)
    assert len(targets[1]) == 0
    with_gt = add_ground_truth_to_proposals(targets, props)
    gi, labels, _, _, rows = heads._match_batch(with_gt, targets)
    per_image = [heads._match_one_image(p, t) for p, t in zip(with_gt, targets)]
    want = torch.cat([m[1] for m in per_image])
    assert torch.equal(labels.cpu(), want.cpu())
    n0 = len(with_gt[0])
    seg = labels[n0:n0 + len(with_gt[1])]
    assert bool((seg == 80).all())                                            # background, none ignored
    assert int((labels == -1).sum()) > 0 and int((labels == 80).sum()) > len(seg)      # the other images use all three intervals
    rows = rows.cpu()
    assert int(rows[1, 0]) == 0 and int(rows[1, 1]) == len(with_gt[1])


def test_image_without_ground_truth_is_background_under_an_ignore_matcher_cpu(oracle):
    _three_interval_check(oracle, "cpu")


@pytest.mark.gpu
def test_image_without_ground_truth_is_background_under_an_ignore_matcher_gpu(oracle):
    _three_interval_check(oracle, "cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [True, False], ids=["sampling_kernel", "torch_ops"])
def test_speculated_sample_equals_the_host_driven_one(oracle, kernel, monkeypatch):
    """The training forwards form the sample on the device WITHOUT the host read, assuming every image fills its budget
    (_label_speculate: locov_sample_proposals, or with LOCOV_LABEL_SAMPLE_KERNEL=0 the torch-op form behind two global sorts), and
    validate afterwards (_label_validate).  On the same draw the speculated Instances equal the host-driven ones field for field,
    in the same field order; a batch that cannot fill its budget (too few background candidates) fails the validation, and the
    forward then returns the reference's counts."""
    from locov_amd.roi_heads import labelling as roi_emb_heads
    from locov_amd.roi_heads.roi_emb_heads import get_event_storage
    monkeypatch.setattr(roi_emb_heads, "_SAMPLE_KERNEL", kernel)
    heads = _heads(True, 64, 0.25, "cuda")
    rng = np.random.default_rng(33)
    props, targets, _ = _batch(oracle, rng, "cuda", n_img=3, r=150, n_gt=6)
    torch.manual_seed(4)
    st_a = heads._label_begin(props, targets)
    torch.manual_seed(4)
    st_b = heads._label_begin(props, targets)
    spec = heads._label_speculate(st_a)
    assert ("lean" in st_a) == kernel and (st_a.get("rois") is not None) == kernel
    assert spec is not None and heads._label_validate(st_a)
    want = heads._label_finish(st_b)
    assert len(spec) == len(want) == 3
    if kernel:          # the pooler's rows come out of the same launch
        from locov_amd.poolers import convert_boxes_to_pooler_format
        assert torch.equal(heads._sampled_rois(st_a, spec), convert_boxes_to_pooler_format([x.proposal_boxes for x in want]))
    for a, b in zip(spec, want):
        assert len(a) == len(b) == 64 and list(a.get_fields()) == list(b.get_fields())
        for k in b.get_fields():
            va, vb = a.get(k), b.get(k)
            assert torch.equal(va.tensor if hasattr(va, "tensor") else va, vb.tensor if hasattr(vb, "tensor") else vb), k
    assert get_event_storage().scalars["roi_head/num_fg_samples"] > 0
    # too few background candidates: every proposal sits on a ground-truth box, 25 % of a budget of 64 may be foreground
    for p, t in zip(props, targets):
        if len(t):
            g = t.gt_boxes.tensor
            p.proposal_boxes = Boxes(g[torch.arange(len(p)) % len(g)] + torch.rand(len(p), 4, device="cuda") * 2 - 1)
    st_c = heads._label_begin(props, targets)
    spec = heads._label_speculate(st_c)
    assert spec is not None and not heads._label_validate(st_c)
    true = heads._label_finish(st_c)
    assert [len(x) for x in true] != [64, 64, 64] and all(len(x) <= 64 for x in true)
    feat = torch.randn(3, 128, 50, 84, device="cuda", requires_grad=True)
    _, box_feats, sampled, losses = heads(None, {"res4": feat}, props, targets)
    assert [len(x) for x in sampled] == [b.shape[0] for b in box_feats] and all(len(x) <= 64 for x in sampled)
    assert [len(x) for x in sampled][1] == 64                       # (the image without ground truth: background only, budget filled)
    assert all(bool(torch.isfinite(v)) for v in losses.values())


@pytest.mark.gpu
@pytest.mark.parametrize("append_gt,logits,n_img,r", [(True, True, 1, 64), (False, True, 4, 64), (True, False, 5, 150), (False, False, 2, 97),
                                                      (True, True, 3, 4090), (True, True, 2, 5000)])
def test_lean_labelling_edge_cases_equal_the_host_driven_form(oracle, append_gt, logits, n_img, r):
    """The lean front end (one concatenation per field, labelling kernel, sampling kernel) on the shapes at its borders: a single
    image; exactly the budget of candidates; ground truth not appended; proposals without objectness_logits; 4 096 candidates per
    image with the appended ground truth; more than that (the general path must take over) -- always the host-driven Instances."""
    heads = _heads(True, 64, 0.25, "cuda")
    heads.proposal_append_gt = append_gt
    rng = np.random.default_rng(50 + n_img + r)
    props, targets, _ = _batch(oracle, rng, "cuda", n_img=n_img, r=r, n_gt=6)
    if not logits:
        for p in props:
            p.remove("objectness_logits")
    torch.manual_seed(9)
    st_a = heads._label_begin(props, targets)
    torch.manual_seed(9)
    st_b = heads._label_begin(props, targets)
    assert ("lean" in st_a) == (r + (6 if append_gt else 0) <= 4096)
    spec = heads._label_speculate(st_a)
    assert spec is not None
    if heads._label_validate(st_a):
        want = heads._label_finish(st_b)
        for a, b in zip(spec, want):
            assert len(a) == len(b) == 64 and list(a.get_fields()) == list(b.get_fields())
            for k in b.get_fields():
                va, vb = a.get(k), b.get(k)
                assert torch.equal(va.tensor if hasattr(va, "tensor") else va, vb.tensor if hasattr(vb, "tensor") else vb), k
    else:               # (exactly the budget of candidates and some of them ignored: the batch goes the host-driven way)
        assert r == 64
        assert all(len(x) <= 64 for x in heads._label_finish(st_a))


@pytest.mark.gpu
def test_sampling_kernel_equals_the_two_global_sorts():
    """locov_sample_proposals against what it replaces -- argsort of the two image-major keys, then per image the first num_pos
    entries of the foreground order and budget - num_pos of the background order -- on images of 16 .. 4 096 proposals (powers of
    two, one above, one below), budgets that are / are not filled, duplicate keys (ties go to the lower row), an image without
    ground truth; every gathered field equals the indexed tensor."""
    from locov_amd import ops
    gen = torch.Generator().manual_seed(12)
    n_r = [16, 1000, 1025, 4096, 64, 333, 2047]
    n_g = [3, 7, 0, 5, 1, 9, 2]
    B, max_pos, K = 16, 4, 80
    total, tg = sum(n_r), sum(n_g)
    off_r, off_g = np.concatenate([[0], np.cumsum(n_r)]), np.concatenate([[0], np.cumsum(n_g)])
    img = torch.repeat_interleave(torch.arange(len(n_r)), torch.tensor(n_r))
    # labels: a few foreground classes, background K, ignored -1; the image without ground truth: all background
    labels = torch.where(torch.rand(total, generator=gen) < 0.02, torch.randint(0, K, (total,), generator=gen), torch.full((total,), K))
    labels[torch.rand(total, generator=gen) < 0.05] = -1
    labels[img == 2] = K
    labels[off_r[4]:off_r[5]] = K                                  # image 4: no foreground at all
    labels[off_r[5]:off_r[5] + 200] = 3                            # image 5: more foreground than max_pos
    labels[off_r[0]:off_r[0] + 14] = 5                             # image 0 (16 proposals): 2 background candidates -> budget NOT filled
    pos, neg = (labels != -1) & (labels != K), labels == K
    rnd = torch.rand((2, total), generator=gen, dtype=torch.float64)
    rnd[0, off_r[1] + 10:off_r[1] + 20] = rnd[0, off_r[1] + 10]    # ties
    rnd[1, off_r[3] + 100:off_r[3] + 140] = 0.0
    key_pos = rnd[0] + (~pos).double() * 2.0 + img.double() * 4.0
    key_neg = rnd[1] + (~neg).double() * 2.0 + img.double() * 4.0
    rows = torch.zeros((len(n_r), 4), dtype=torch.int64)
    for i in range(len(n_r)):
        sl = slice(off_r[i], off_r[i + 1])
        rows[i, 0], rows[i, 1] = int(pos[sl].sum()), int(neg[sl].sum())
    gt_index = torch.zeros(total, dtype=torch.int64)
    for i in range(len(n_r)):
        if n_g[i]:
            gt_index[off_r[i]:off_r[i + 1]] = torch.randint(int(off_g[i]), int(off_g[i + 1]), (n_r[i],), generator=gen)
    boxes = torch.rand((total, 4), generator=gen) * 500
    gtb = torch.rand((tg, 4), generator=gen) * 500
    field = torch.randn(total, generator=gen)
    d = lambda t: t.cuda()
    picked, ob, oc, og, fg, rois, fo = ops.sample_proposals(d(key_pos), d(key_neg), d(labels), d(gt_index), d(rows), d(boxes), d(gtb), n_r, n_g,
                                                            B, max_pos, K, field=d(field))
    torch.cuda.synchronize()
    # expectation: stable sorts (ties -> lower row), per image
    want = []
    for i in range(len(n_r)):
        sl = slice(int(off_r[i]), int(off_r[i + 1]))
        po = np.argsort(key_pos[sl].numpy(), kind="stable") + off_r[i]
        no = np.argsort(key_neg[sl].numpy(), kind="stable") + off_r[i]
        num_pos = min(int(rows[i, 0]), max_pos)
        take = lambda order, k: [int(order[min(j, n_r[i] - 1)]) for j in range(k)]
        want += take(po, num_pos) + take(no, B - num_pos)
    want = torch.tensor(want)
    assert torch.equal(picked.cpu(), want)
    filled = [min(int(rows[i, 0]), max_pos) + int(rows[i, 1]) >= B for i in range(len(n_r))]
    assert filled == [False, True, True, True, True, True, True]
    for i in range(len(n_r)):
        if filled[i]:           # a filled budget holds foreground, then background -- never an ignored proposal
            c = oc[i * B:(i + 1) * B].cpu()
            k = min(int(rows[i, 0]), max_pos)
            assert bool(((c[:k] >= 0) & (c[:k] < K)).all()) and bool((c[k:] == K).all())
    # classes: labels[picked] -- except in the slots of an image that does NOT fill its budget which read past the background
    # population: the caller throws that sample away but has already enqueued the losses on it, so such a slot counts as
    # background (never the ignore label -1, which torch's cross-entropy answers with a device-side assert)
    want_cls = labels[want].clone()
    for i in range(len(n_r)):
        k = min(int(rows[i, 0]), max_pos)
        past = k + int(rows[i, 1])
        if past < B:
            want_cls[i * B + past:(i + 1) * B] = K
    assert bool((oc.cpu() != -1).all())
    assert torch.equal(ob.cpu(), boxes[want]) and torch.equal(oc.cpu(), want_cls) and torch.equal(fo.cpu(), field[want])
    assert torch.equal(fg.cpu(), (want_cls != K).long())
    has_gt = torch.tensor([n_g[i] > 0 for i in range(len(n_r))]).repeat_interleave(B)
    assert torch.equal(og.cpu()[has_gt], gtb[gt_index[want]][has_gt]) and not bool(og.cpu()[~has_gt].any())
    assert torch.equal(rois.cpu(), torch.cat([torch.arange(len(n_r)).repeat_interleave(B).float()[:, None], boxes[want]], dim=1))
    # without the extra field, and the argument checks of the wrapper
    out = ops.sample_proposals(d(key_pos), d(key_neg), d(labels), d(gt_index), d(rows), d(boxes), d(gtb), n_r, n_g, B, max_pos, K)
    assert out[-1] is None and torch.equal(out[0].cpu(), want)
    with pytest.raises(ValueError):
        ops.sample_proposals(d(key_pos), d(key_neg), d(labels), d(gt_index), d(rows), d(boxes), d(gtb), n_r[:-1] + [5000], n_g, B, max_pos, K)


@pytest.mark.gpu
def test_training_forward_with_the_sampling_kernel_equals_the_torch_op_form(oracle, monkeypatch):
    """One training forward + backward of the LSM heads with the lean labelling + sampling kernel and with the torch-op form, same
    seeds: the same sampled proposals, losses and Res5 weight gradients, bit for bit."""
    from locov_amd.roi_heads import labelling as roi_emb_heads
    outs = {}
    for kernel in (True, False):
        monkeypatch.setattr(roi_emb_heads, "_SAMPLE_KERNEL", kernel)
        heads = _heads(True, 64, 0.25, "cuda")
        rng = np.random.default_rng(33)
        props, targets, _ = _batch(oracle, rng, "cuda", n_img=3, r=150, n_gt=6)
        feat = torch.randn(3, 128, 50, 84, generator=torch.Generator().manual_seed(2)).cuda().requires_grad_(True)
        torch.manual_seed(4)
        grid, box_feats, sampled, losses = heads(None, {"res4": feat}, props, targets)
        (sum(losses.values()) + 1e-3 * grid.square().mean() + 1e-2 * torch.cat(box_feats).square().mean()).backward()
        outs[kernel] = ([x.proposal_boxes.tensor.clone() for x in sampled], [x.gt_classes.clone() for x in sampled],
                        {k: float(v) for k, v in losses.items()},
                        {k: p.grad.clone() for k, p in heads.named_parameters() if p.grad is not None and k.startswith("res5.")})
    a, b = outs[True], outs[False]
    assert all(torch.equal(x, y) for x, y in zip(a[0], b[0])) and all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
    assert a[2] == b[2] and set(a[3]) == set(b[3]) and len(a[3]) >= 10
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k


@pytest.mark.gpu
def test_label_kernel_equals_the_torch_ops_on_the_device(oracle, monkeypatch):
    """locov_label_proposals (one launch per batch) against the torch-op form of SampleAllROIHeads._match_batch on the same draw:
    matched ground truth, labels, both sampling orders and the per-image rows -- equal element for element, incl. an image without
    ground truth, a NaN proposal, a zero-width foreground candidate and exact ties."""
    from locov_amd import ops
    heads = _heads(True, 64, 0.5, "cuda")
    rng = np.random.default_rng(21)
    props, targets, raw = _batch(oracle, rng, "cuda", n_img=5, r=300, n_gt=7)
    b0 = props[0].proposal_boxes.tensor.clone()
    b0[11, 1] = float("nan")                                                  # NaN coordinate: IoU 0 by the `inter > 0` rule
    b0[12] = targets[0].gt_boxes.tensor[2]                                    # an exact copy of a ground-truth box: IoU 1
    b0[13] = targets[0].gt_boxes.tensor[2]
    b0[13, 2] = b0[13, 0]                                                     # ... and a zero-width one
    b0[14] = torch.tensor([0.0, 0.0, 1.0, 1.0])                               # touches nothing: every IoU is 0 (a tie: first row wins)
    props[0].proposal_boxes = Boxes(b0)
    from locov_amd.roi_heads.roi_emb_heads import add_ground_truth_to_proposals
    props = add_ground_truth_to_proposals(targets, props)
    outs = []
    for use_kernel in (True, False):
        monkeypatch.setattr(ops, "LABEL_MAX_IMAGES", 64 if use_kernel else 0)
        torch.manual_seed(9)
        outs.append(heads._match_batch(props, targets))
    (gi_k, lab_k, po_k, no_k, rows_k), (gi_t, lab_t, po_t, no_t, rows_t) = outs
    assert torch.equal(lab_k, lab_t) and torch.equal(gi_k, gi_t)
    assert torch.equal(po_k, po_t) and torch.equal(no_k, no_t)
    assert torch.equal(rows_k[:, :2], rows_t[:, :2])
    assert torch.equal(rows_k[:, 2:] > 0, rows_t[:, 2:] > 0)
    assert int((lab_k == 80).sum()) > 0 and int(((lab_k >= 0) & (lab_k < 80)).sum()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("agnostic", [True, False])
@pytest.mark.parametrize("beta", [0.0, 0.5])
def test_one_launch_box_reg_loss_equals_the_torch_chain(agnostic, beta):
    """ops.box_reg_loss (locov_box_reg_loss: deltas, smooth-L1, masked sum, normalisation and the gradient in one launch) against the
    torch-op chain of box_reg_loss it replaces when the labelling has validated the boxes: value and gradient, class-agnostic and
    per-class predictions, L1 and smooth-L1, non-finite predictions in background / ignored rows."""
    heads = _heads(False, 16, 0.5, "cuda")
    bp = heads.box_predictor
    bp.smooth_l1_beta = beta
    g = torch.Generator().manual_seed(3)
    n, K = 700, bp.num_classes
    boxes = torch.rand(n, 4, generator=g) * 600
    boxes[:, 2:] = boxes[:, :2] + 4 + torch.rand(n, 2, generator=g) * 300
    gt = boxes + torch.randn(n, 4, generator=g) * 6
    gt[:, 2:] = torch.maximum(gt[:, 2:], gt[:, :2] + 1)
    cls = torch.randint(0, K, (n,), generator=g)
    cls[torch.rand(n, generator=g) < 0.5] = K                       # background
    cls[torch.rand(n, generator=g) < 0.05] = -1                     # ignored
    pred = torch.randn(n, 4 if agnostic else 4 * K, generator=g) * 0.3
    bad = (cls == K).nonzero()[:3, 0]
    pred[bad[0]] = float("inf"); pred[bad[1]] = float("nan"); pred[bad[2], 1] = -float("inf")
    boxes, gt, cls = boxes.cuda(), gt.cuda(), cls.cuda()
    a = pred.clone().cuda().requires_grad_(True)
    b = pred.clone().cuda().requires_grad_(True)
    want = bp.box_reg_loss(boxes, gt, a, cls, boxes_validated=False)          # the torch-op chain
    got = bp.box_reg_loss(boxes, gt, b, cls, boxes_validated=True)            # one launch
    assert got.shape == want.shape and torch.isfinite(got)
    (want * 3.0).backward()
    (got * 3.0).backward()
    assert abs(float(got) - float(want)) <= 2e-6 * abs(float(want))
    assert torch.isfinite(b.grad).all()
    torch.testing.assert_close(b.grad, a.grad, rtol=1e-6, atol=1e-9)
    assert torch.equal(b.grad != 0, a.grad != 0)                               # exactly the foreground rows' (class) columns
    # an all-background batch: zero loss, zero gradient
    c = pred.clone().cuda().requires_grad_(True)
    z = bp.box_reg_loss(boxes, gt, c, torch.full_like(cls, K), boxes_validated=True)
    z.backward()
    assert float(z) == 0.0 and not c.grad.any()
