"""Edge cases of the plugin-level path on the GPU (SURVEY.md section 3 "cover the edge cases the
reference tests": empty and ragged inputs, normalised / standardised classifier, odd sizes)."""
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


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _cfg(pkg, **box_head):
    cfg = pkg.config.get_cfg()
    cfg.MODEL.RESNETS.RES2_OUT_CHANNELS = 32
    cfg.MODEL.RESNETS.WIDTH_PER_GROUP = 8
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    cfg.MODEL.ROI_BOX_HEAD.EMB_DIM = 96
    cfg.MODEL.ROI_HEADS.NAME = "EmbeddingProposalsRes5ROIHeads"
    for k, v in box_head.items():
        cfg.MODEL.ROI_BOX_HEAD[k] = v
    return cfg


def _heads(pkg, oracle, cfg, k=17, seed=3):
    from locov_amd.structures import ShapeSpec
    heads = pkg.build_roi_heads(cfg, {"res4": ShapeSpec(channels=128, stride=16)})
    params = oracle.make_res5_params(seed, in_ch=128, mid=64, out_ch=256)
    heads.res5.load_state_dict(params)
    h = oracle.synth_head(np.random.default_rng(seed), 256, 96, k)
    bp = heads.box_predictor
    with torch.no_grad():
        bp.emb_pred.weight.copy_(torch.from_numpy(h["emb_w"]))
        bp.emb_pred.bias.copy_(torch.from_numpy(h["emb_b"]) + 0.01)
        h["emb_b"] = h["emb_b"] + np.float32(0.01)
        bp.bbox_pred.weight.copy_(torch.from_numpy(h["bbox_w"]))
        bp.bbox_pred.bias.copy_(torch.from_numpy(h["bbox_b"]))
    heads = heads.cuda().eval()
    bp.set_class_embeddings(h["cls_w"])
    heads.num_classes = bp.num_classes
    return heads, params, h


def _props(boxes_list):
    from locov_amd.structures import Boxes, Instances
    out = []
    for b in boxes_list:
        inst = Instances((800, 1333))
        inst.proposal_boxes = Boxes(torch.from_numpy(b).cuda())
        inst.objectness_logits = torch.zeros(len(b), device="cuda")
        out.append(inst)
    return out


@pytest.mark.parametrize("norm,std", [(True, False), (False, True)])
def test_normalised_and_standardised_classifier(pkg, oracle, norm, std):
    """NORMALIZE_EMB_PRED / STANDARDIZE_EMB_PRED (box_emb_head.py:207-210 on the embeddings, :223-232
    on the bank)."""
    cfg = _cfg(pkg, NORMALIZE_EMB_PRED=norm, STANDARDIZE_EMB_PRED=std)
    heads, params, h = _heads(pkg, oracle, cfg)
    rng = np.random.default_rng(5)
    feat = rng.standard_normal((2, 128, 50, 84)).astype(np.float32)
    boxes = [oracle.synth_boxes(rng, 33), oracle.synth_boxes(rng, 20)]
    cls_w, _, _ = oracle.set_class_embeddings(h["cls_w"], norm, std)
    np.testing.assert_allclose(heads.box_predictor.cls_score.weight.cpu().numpy(), cls_w, rtol=2e-5, atol=2e-6)
    want = oracle.roi_head_forward(feat, boxes, params, dict(h, cls_w=cls_w), normalize_emb=norm, standardize_emb=std)
    with torch.no_grad():
        bf = heads._shared_roi_transform([dev(feat)], [p.proposal_boxes for p in _props(boxes)])
        scores, deltas = heads.box_predictor(heads._pooled_mean(bf))
    np.testing.assert_allclose(scores.cpu().numpy(), want["scores"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(deltas.cpu().numpy(), want["deltas"], atol=1e-5)


def test_ragged_and_empty_images(pkg, oracle):
    """One image without proposals, one with a single proposal, one with 130 (not a tile multiple)."""
    heads, params, h = _heads(pkg, oracle, _cfg(pkg))
    rng = np.random.default_rng(6)
    feat = rng.standard_normal((3, 128, 50, 84)).astype(np.float32)
    boxes = [np.zeros((0, 4), np.float32), oracle.synth_boxes(rng, 1), oracle.synth_boxes(rng, 130)]
    want = oracle.roi_head_forward(feat, boxes, params, h)
    props = _props(boxes)
    with torch.no_grad():
        inst, _ = heads(None, {"res4": dev(feat)}, props, None)
        bf = heads._shared_roi_transform([dev(feat)], [p.proposal_boxes for p in props])
        scores, deltas = heads.box_predictor(heads._pooled_mean(bf))
    assert len(inst) == 3 and len(inst[0]) == 0
    assert tuple(bf.shape) == (131, 256, 7, 7)
    np.testing.assert_allclose(scores.cpu().numpy(), want["scores"], atol=1e-4)
    np.testing.assert_allclose(deltas.cpu().numpy(), want["deltas"], atol=1e-5)


def test_no_proposals_at_all(pkg, oracle):
    heads, _, _ = _heads(pkg, oracle, _cfg(pkg))
    feat = torch.randn(2, 128, 50, 84, device="cuda")
    props = _props([np.zeros((0, 4), np.float32)] * 2)
    with torch.no_grad():
        inst, losses = heads(None, {"res4": feat}, props, None)
    assert losses == {} and [len(i) for i in inst] == [0, 0]


def test_miopen_and_hip_backends_agree(pkg, oracle):
    rng = np.random.default_rng(8)
    feat = dev(rng.standard_normal((2, 128, 50, 84)).astype(np.float32))
    boxes = [oracle.synth_boxes(rng, 70), oracle.synth_boxes(rng, 45)]
    outs = []
    for backend in ("hip", "miopen"):
        heads, _, _ = _heads(pkg, oracle, _cfg(pkg, RES5_BACKEND=backend))
        with torch.no_grad():
            bf = heads._shared_roi_transform([feat], [p.proposal_boxes for p in _props(boxes)])
            outs.append(heads.box_predictor(heads._pooled_mean(bf))[0].cpu().numpy())
    np.testing.assert_allclose(outs[0], outs[1], atol=2e-4)


def test_bf16_similarity_mode_through_the_predictor(pkg, oracle):
    heads, params, h = _heads(pkg, oracle, _cfg(pkg, SIM_GEMM_DTYPE="bf16"), k=1203)
    rng = np.random.default_rng(9)
    x = np.maximum(rng.standard_normal((257, 256)), 0).astype(np.float32)
    scores32, _, emb = oracle.box_predictor_forward(x, h["emb_w"], h["emb_b"], h["bbox_w"], h["bbox_b"], h["cls_w"])
    with torch.no_grad():
        scores, _ = heads.box_predictor(dev(x))
    want = (torch.from_numpy(emb).to(torch.bfloat16).double() @ torch.from_numpy(h["cls_w"]).to(torch.bfloat16).double().t())
    assert (scores.cpu().double() - want).abs().max().item() <= 2e-4      # fp64 on the same bf16-rounded operands
    assert np.abs(scores.cpu().numpy() - scores32).max() < 5e-2            # vs pure fp32: bf16 input rounding
    assert torch.all(scores[:, -1] == 0)


def test_whole_head_replays_from_a_hip_graph(pkg, oracle):
    """The C ABI only enqueues work on the caller's stream (no allocation, no sync, no host-side state):
    after one eager warm-up call (which sizes the cached workspaces) the complete fused head -- layout
    change, ROIAlign, Res5 (Winograd + batched GEMM), mean, FCs, similarity GEMM -- is captured into a
    HIP graph and replayed on new inputs."""
    from locov_amd import ops
    heads, params, h = _heads(pkg, oracle, _cfg(pkg))
    rng = np.random.default_rng(12)
    feat = dev(rng.standard_normal((2, 128, 50, 84)).astype(np.float32))
    boxes = [oracle.synth_boxes(rng, 40), oracle.synth_boxes(rng, 40)]
    rois = torch.cat([torch.cat([torch.full((40, 1), float(i)), torch.from_numpy(b)], 1) for i, b in enumerate(boxes)]).cuda()
    bp = heads.box_predictor

    def run(f, r):
        nhwc = ops.nchw_to_nhwc(f)
        x0 = heads.res5.rows_input(49 * r.shape[0], f.device)
        ops.roi_align_nhwc(nhwc, r, 14, 1.0 / 16, 0, True, bin_stride=2, pos_major=True, out=x0)
        y = heads.res5.forward_rows(x0, 7, 7, pos_major=True)
        return ops.box_head(y.view(7, 7, r.shape[0], -1), bp.emb_pred.weight, bp.emb_pred.bias, bp.bbox_pred.weight,
                            bp.bbox_pred.bias, bp.cls_score.weight, None, ops.NORM_NONE, ops.F32, channels_last=2)

    with torch.no_grad():
        eager = [t.clone() for t in run(feat, rois)]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            run(feat, rois)                                   # warm-up on the capture stream
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = run(feat, rois)
        # new inputs written into the captured buffers, then replay
        feat2 = dev(rng.standard_normal((2, 128, 50, 84)).astype(np.float32))
        want2 = [t.clone() for t in run(feat2, rois)]
        feat.copy_(feat2)
        g.replay()
        torch.cuda.synchronize()
    assert not torch.equal(eager[3], want2[3])
    assert all(torch.equal(a, b) for a, b in zip(out, want2))        # pooled, deltas, embeddings, logits


def test_f32_gemm_random_shapes(pkg):
    """40 random (M, N, K, strided lda, epilogue) combinations of the f32-MFMA NT GEMM against fp64 (ragged tiles, K not a
    multiple of the K-tile, every tile configuration the dispatcher picks)."""
    ops = pkg.ops
    rng = np.random.default_rng(4242)
    for _ in range(40):
        M = int(rng.integers(1, 700))
        N = int(rng.integers(1, 400))
        K = 4 * int(rng.integers(1, 80))
        pad = 4 * int(rng.integers(0, 5))
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        wide = torch.randn(M, K + pad, generator=g).cuda()
        x = wide[:, :K] if pad else wide
        w = (torch.randn(N, K, generator=g) * 0.05).cuda()
        use_res, use_aff, relu = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2))
        res = torch.randn(M, N, generator=g).cuda() if use_res else None
        sc = (torch.rand(N, generator=g) + 0.5).cuda() if use_aff else None
        sh = torch.randn(N, generator=g).cuda() if use_aff else None
        ref = x.double() @ w.double().t()
        if use_aff:
            ref = ref * sc.double() + sh.double()
        if use_res:
            ref = ref + res.double()
        if relu:
            ref = torch.relu(ref)
        got = ops.linear(x, w, sh, scale=sc, residual=res, relu=relu)
        err = (got.double() - ref).abs().max().item()
        assert err <= 4e-6 * max(1.0, float(ref.abs().max())), (M, N, K, pad, use_res, use_aff, relu, err)


def test_batched_postprocess_equals_the_per_image_form():
    """fast_rcnn_inference (roi_emb_heads.py:280,357) takes the whole batch through one set of device ops; it must return exactly
    what fast_rcnn_inference_single_image returns image by image -- ragged batch, different image sizes, an image without
    proposals, one without a candidate above the threshold, duplicates (NMS), more survivors than top-k."""
    from locov_amd.roi_heads import box_emb_head as beh
    g = torch.Generator().manual_seed(11)
    sizes, shapes = [120, 0, 300, 40], [(800, 1333), (640, 480), (427, 640), (800, 1200)]
    K = 17
    boxes, scores = [], []
    for n, (h, w) in zip(sizes, shapes):
        xy = torch.rand(n, 2, generator=g) * torch.tensor([w * 1.1, h * 1.1]) - 20           # some boxes leave the image
        b = torch.cat([xy, xy + 10 + torch.rand(n, 2, generator=g) * 200], 1)
        if n >= 100:
            b[10:40] = b[:30] + torch.rand(30, 4, generator=g) * 3                             # near-duplicates for the NMS
        s = torch.softmax(torch.randn(n, K + 1, generator=g) * 3, dim=1)
        boxes.append(b.cuda())
        scores.append(s.cuda())
    scores[3] = scores[3] * 1e-3                                                               # nothing above the threshold
    for thresh, topk in ((0.05, 100), (0.2, 5), (0.0, 20)):
        got, got_rows = beh.fast_rcnn_inference(boxes, scores, shapes, thresh, 0.5, topk)
        for i in range(len(sizes)):
            want, want_rows = beh.fast_rcnn_inference_single_image(boxes[i], scores[i], shapes[i], thresh, 0.5, topk)
            assert len(got[i]) == len(want), (thresh, topk, i, len(got[i]), len(want))
            assert torch.equal(got[i].pred_boxes.tensor, want.pred_boxes.tensor)
            assert torch.equal(got[i].scores, want.scores) and torch.equal(got[i].pred_classes, want.pred_classes)
            assert torch.equal(got_rows[i], want_rows)
        assert len(got[1]) == 0 and len(got[3]) == (0 if thresh > 0 else min(topk, len(got[3])))
