"""GPU parity of the plugin-level path: the Detectron2-shaped ROI heads / box predictor of
locov_amd against the CPU oracle on the same seeded inputs.
Gate: fp32 logits within 1e-4 of the oracle (north_star)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

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


def test_conv3x3_nhwc_vs_torch_cpu(pkg):
    ops = pkg.ops
    g = torch.Generator().manual_seed(3)
    for (R, H, W, Cin, N) in ((37, 7, 7, 64, 96), (3, 25, 42, 32, 40), (130, 7, 7, 128, 128)):
        x = torch.randn(R, Cin, H, W, generator=g)
        w = torch.randn(N, Cin, 3, 3, generator=g) * 0.05
        sc = torch.rand(N, generator=g) + 0.5
        sh = torch.randn(N, generator=g)
        want = F.relu(F.conv2d(x.double(), w.double(), padding=1) * sc.double().view(1, -1, 1, 1)
                      + sh.double().view(1, -1, 1, 1))
        rows = x.permute(0, 2, 3, 1).reshape(R * H * W, Cin).contiguous().cuda()
        wp = ops.pack_conv3x3_weight(w.cuda())
        np.testing.assert_array_equal(wp.cpu().numpy(), w.permute(0, 2, 3, 1).reshape(N, 9 * Cin).numpy())
        got = ops.conv3x3_nhwc(rows, wp, H, W, scale=sc.cuda(), shift=sh.cuda(), relu=True)
        got = got.view(R, H, W, N).permute(0, 3, 1, 2).cpu().double()
        assert (got - want).abs().max().item() < 2e-5
        # position-major rows [H,W,R,C]: same convolution, zero-padding taps skipped instead of multiplied
        rows_p = x.permute(2, 3, 0, 1).reshape(H * W * R, Cin).contiguous().cuda()
        res_p = torch.randn(H * W * R, N, generator=g)
        got = ops.conv3x3_nhwc(rows_p, wp, H, W, scale=sc.cuda(), shift=sh.cuda(), residual=res_p.cuda(),
                               relu=True, pos_major=True)
        want_p = F.relu(F.conv2d(x.double(), w.double(), padding=1) * sc.double().view(1, -1, 1, 1)
                        + sh.double().view(1, -1, 1, 1) + res_p.double().view(H, W, R, N).permute(2, 3, 0, 1))
        got = got.view(H, W, R, N).permute(2, 3, 0, 1).cpu().double()
        assert (got - want_p).abs().max().item() < 2e-5


def _small_cfg(pkg):
    cfg = pkg.config.get_cfg()
    cfg.MODEL.RESNETS.RES2_OUT_CHANNELS = 32        # res5: 128 -> (64) -> 256
    cfg.MODEL.RESNETS.WIDTH_PER_GROUP = 8
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    cfg.MODEL.ROI_BOX_HEAD.EMB_DIM = 96
    cfg.MODEL.ROI_HEADS.NAME = "EmbeddingProposalsRes5ROIHeads"
    return cfg


def test_res5_rows_path_vs_oracle(pkg, oracle):
    from locov_amd.res5 import build_res5_block
    res5, out_ch = build_res5_block(_small_cfg(pkg))
    params = oracle.make_res5_params(5, in_ch=128, mid=64, out_ch=256)
    res5.load_state_dict(params)
    res5 = res5.cuda().eval()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(21, 128, 14, 14, generator=g)
    want = oracle.res5_stage(x, params).numpy()
    # stock path (torch conv2d on the GPU = MIOpen)
    with torch.no_grad():
        got_m = res5(x.cuda()).cpu().numpy()
    np.testing.assert_allclose(got_m, want, atol=2e-4, rtol=1e-4)
    # hand-written rows path on the even positions
    x0 = x[:, :, ::2, ::2].permute(0, 2, 3, 1).reshape(21 * 49, 128).contiguous().cuda()
    got = res5.forward_rows(x0, 7, 7, winograd=False).view(21, 7, 7, out_ch).permute(0, 3, 1, 2).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-5)
    # (ROI-major rows also go through the Winograd form, the default: bounded relative to the activation range)
    got = res5.forward_rows(x0, 7, 7).view(21, 7, 7, out_ch).permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()
    # same, position-major rows
    x0p = x[:, :, ::2, ::2].permute(2, 3, 0, 1).reshape(49 * 21, 128).contiguous().cuda()
    got = res5.forward_rows(x0p, 7, 7, pos_major=True, winograd=False).view(7, 7, 21, out_ch).permute(2, 3, 0, 1)
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=2e-5, rtol=1e-5)
    # same, 3x3 convolutions in the Winograd domain (the default): rounding differs, bounded relative to
    # the activation range (the logits gate of 1e-4 is checked in the heads tests below)
    got = res5.forward_rows(x0p, 7, 7, pos_major=True, winograd=True).view(7, 7, 21, out_ch).permute(2, 3, 0, 1)
    assert np.abs(got.cpu().numpy() - want).max() <= 2e-5 * np.abs(want).max()


def test_block0_convolutions_on_the_map_vs_oracle(pkg, oracle):
    """Res5Stage.forward_from_map: conv1 / projection shortcut of block 0 applied to the res4 map and pooled
    afterwards (ROIAlign is linear) -- against the oracle, which pools first like the reference, and against
    this package's pooled-rows path."""
    from locov_amd import ops
    from locov_amd.res5 import build_res5_block
    res5, out_ch = build_res5_block(_small_cfg(pkg))
    params = oracle.make_res5_params(11, in_ch=128, mid=64, out_ch=256)
    res5.load_state_dict(params)
    res5 = res5.cuda().eval()
    rng = np.random.default_rng(21)
    feat = rng.standard_normal((2, 128, 50, 84)).astype(np.float32)
    boxes = [oracle.synth_boxes(rng, 90), oracle.synth_boxes(rng, 75)]
    rois = oracle.boxes_to_pooler_format(boxes)
    pooled = oracle.roi_align(feat, rois, (14, 14), 1.0 / 16, 0, True)
    want = oracle.res5_stage(pooled, params).numpy()                                  # [R,256,7,7]
    R = rois.shape[0]
    assert res5.map_path_pays(1000, 4200) and not res5.map_path_pays(100, 4200)
    with torch.no_grad():
        nhwc = ops.nchw_to_nhwc(dev(feat))
        for wino in (False, True):
            got = res5.forward_from_map(nhwc, dev(rois), 14, 1.0 / 16, 0, True, winograd=wino)
            got = got.view(7, 7, R, out_ch).permute(2, 3, 0, 1).cpu().numpy()
            assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max(), (wino, np.abs(got - want).max())
            x0 = res5.rows_input(49 * R, nhwc.device)
            ops.roi_align_nhwc(nhwc, dev(rois), 14, 1.0 / 16, 0, True, bin_stride=2, pos_major=True, out=x0)
            ref = res5.forward_rows(x0, 7, 7, pos_major=True, winograd=wino)
            ref = ref.view(7, 7, R, out_ch).permute(2, 3, 0, 1).cpu().numpy()
            assert np.abs(got - ref).max() <= 2e-5 * np.abs(want).max()


def _make_heads(pkg, oracle, cfg, k_classes, seed, res5_dims=None):
    from locov_amd.structures import ShapeSpec
    c_in = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS * 4
    heads = pkg.build_roi_heads(cfg, {"res4": ShapeSpec(channels=c_in, stride=16)})
    c5, mid = heads.output_shape, cfg.MODEL.RESNETS.WIDTH_PER_GROUP * 8
    params = oracle.make_res5_params(seed, in_ch=c_in, mid=mid, out_ch=c5)
    heads.res5.load_state_dict(params)
    rng = np.random.default_rng(seed)
    h = oracle.synth_head(rng, c5, cfg.MODEL.ROI_BOX_HEAD.EMB_DIM, k_classes)
    bp = heads.box_predictor
    with torch.no_grad():
        bp.emb_pred.weight.copy_(torch.from_numpy(h["emb_w"]))
        bp.emb_pred.bias.copy_(torch.from_numpy(h["emb_b"]))
        bp.bbox_pred.weight.copy_(torch.from_numpy(h["bbox_w"]))
        bp.bbox_pred.bias.copy_(torch.from_numpy(h["bbox_b"]))
    heads = heads.cuda().eval()
    bp.set_class_embeddings(h["cls_w"])           # trainer.py:365-396 load_embeddings contract
    heads.num_classes = bp.num_classes
    return heads, params, h


def _proposals(pkg, oracle, rng, n_img, r, device="cuda"):
    from locov_amd.structures import Boxes, Instances
    out, boxes = [], []
    for _ in range(n_img):
        b = oracle.synth_boxes(rng, r)
        boxes.append(b)
        inst = Instances((800, 1333))
        inst.proposal_boxes = Boxes(torch.from_numpy(b).to(device))
        inst.objectness_logits = torch.zeros(r, device=device)
        out.append(inst)
    return out, boxes


@pytest.mark.parametrize("backend", ["hip", "hip-winograd", "miopen"])
def test_roi_heads_small_vs_oracle(pkg, oracle, backend):
    cfg = _small_cfg(pkg)
    cfg.MODEL.ROI_BOX_HEAD.RES5_BACKEND = backend.split("-")[0]
    cfg.MODEL.ROI_BOX_HEAD.RES5_CONV3X3 = "winograd" if backend == "hip-winograd" else "direct"
    heads, params, h = _make_heads(pkg, oracle, cfg, 80, 7)
    rng = np.random.default_rng(17)
    feat = rng.standard_normal((2, 128, 50, 84)).astype(np.float32)
    props, boxes = _proposals(pkg, oracle, rng, 2, 60)
    want = oracle.roi_head_forward(feat, boxes, params, h)
    with torch.no_grad():
        bf = heads._shared_roi_transform([dev(feat)], [p.proposal_boxes for p in props])
        assert tuple(bf.shape) == (120, 256, 7, 7)
        pooled = heads._pooled_mean(bf)
        scores, deltas = heads.box_predictor(pooled)
    tol = 2e-5 if backend == "hip" else 2e-4
    np.testing.assert_allclose(bf.cpu().numpy(), want["res5"], atol=tol, rtol=1e-4)
    np.testing.assert_allclose(pooled.cpu().numpy(), want["box_features"], atol=tol, rtol=1e-4)
    np.testing.assert_allclose(scores.cpu().numpy(), want["scores"], atol=1e-4)
    np.testing.assert_allclose(deltas.cpu().numpy(), want["deltas"], atol=1e-5)
    # full inference entry point: same detections as the oracle's post-processing of ITS logits
    with torch.no_grad():
        inst, losses = heads(None, {"res4": dev(feat)}, props, None)
    assert losses == {} and len(inst) == 2
    probs = oracle.softmax(want["scores"])
    for i in range(2):
        sl = slice(i * 60, (i + 1) * 60)
        pb = oracle.apply_deltas(want["deltas"][sl], boxes[i])
        wb, ws, wc = oracle.fast_rcnn_inference_single_image(pb, probs[sl], (800, 1333), 0.05, 0.5, 100)
        assert len(inst[i]) == len(wb)
        np.testing.assert_allclose(inst[i].pred_boxes.tensor.cpu().numpy(), wb, atol=1e-2)
        np.testing.assert_allclose(inst[i].scores.cpu().numpy(), ws, atol=1e-5)
        np.testing.assert_array_equal(inst[i].pred_classes.cpu().numpy(), wc)


@pytest.mark.parametrize("conv3x3", ["winograd", "direct"])
def test_roi_heads_reference_config_vs_oracle(pkg, oracle, conv3x3):
    """configs/coco_lsm.yaml shapes = config 1 of BASELINE.json at its full size: 2 synthetic 1333x800 images x 100
    proposals, res4 [2,1024,50,84], Res5 1024->2048, D=768, 80-class bank."""
    cfg = pkg.config.get_cfg()
    cfg.MODEL.ROI_BOX_HEAD.RES5_CONV3X3 = conv3x3
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    cfg.MODEL.ROI_HEADS.NAME = "EmbeddingProposalsRes5ROIHeads"
    cfg.MODEL.ROI_HEADS.DETACH_CLASS_PREDICTOR = True
    heads, params, h = _make_heads(pkg, oracle, cfg, 80, 1992)
    rng = np.random.default_rng(1992)
    feat = rng.standard_normal((2, 1024, 50, 84)).astype(np.float32)
    props, boxes = _proposals(pkg, oracle, rng, 2, 100)
    want = oracle.roi_head_forward(feat, boxes, params, h)
    with torch.no_grad():
        bf = heads._shared_roi_transform([dev(feat)], [p.proposal_boxes for p in props])
        scores, deltas = heads.box_predictor(heads._pooled_mean(bf))
    err = np.abs(scores.cpu().numpy() - want["scores"]).max()
    assert err <= 1e-4, err                                  # north_star gate
    np.testing.assert_allclose(deltas.cpu().numpy(), want["deltas"], atol=1e-5)
    assert np.all(scores.cpu().numpy()[:, -1] == 0)          # zero background row


def test_heads_take_the_map_path_with_many_proposals(pkg, oracle):
    """Many proposals per image (49 R >= 3 N H W): EmbeddingProposalsRes5ROIHeads runs block 0's 1x1 convolutions
    on the map (Res5Stage.forward_from_map); logits gate 1e-4 against the oracle, which pools first."""
    from locov_amd.structures import Boxes, Instances
    cfg = _small_cfg(pkg)
    heads, params, h = _make_heads(pkg, oracle, cfg, 80, 31)
    rng = np.random.default_rng(31)
    feat = rng.standard_normal((2, 128, 20, 30)).astype(np.float32)              # a 480 x 320 image at stride 16
    boxes = [oracle.synth_boxes(rng, 300, 480.0, 320.0), oracle.synth_boxes(rng, 260, 480.0, 320.0)]
    assert heads.res5.map_path_pays(560, 2 * 20 * 30)
    calls = []
    orig = heads.res5.forward_from_map
    heads.res5.forward_from_map = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    props = []
    for b in boxes:
        inst = Instances((320, 480))
        inst.proposal_boxes = Boxes(torch.from_numpy(b).cuda())
        inst.objectness_logits = torch.zeros(len(b), device="cuda")
        props.append(inst)
    want = oracle.roi_head_forward(feat, boxes, params, h)
    with torch.no_grad():
        bf = heads._shared_roi_transform([dev(feat)], [p.proposal_boxes for p in props])
        scores, deltas = heads.box_predictor(heads._pooled_mean(bf))
    assert calls, "the map path was not taken"
    assert tuple(bf.shape) == (560, 256, 7, 7)
    assert np.abs(bf.cpu().numpy() - want["res5"]).max() <= 2e-5 * np.abs(want["res5"]).max()
    assert np.abs(scores.cpu().numpy() - want["scores"]).max() <= 1e-4
    np.testing.assert_allclose(deltas.cpu().numpy(), want["deltas"], atol=1e-5)


@pytest.mark.parametrize("many", [False, True])
def test_opt_in_bf16_res5_is_bounded_against_the_fp32_path(pkg, oracle, many):
    """MODEL.ROI_BOX_HEAD.RES5_DTYPE = "bf16" (extension, default "fp32"): bf16 GEMM operands, fp32 accumulate and
    epilogues. It is NOT a parity configuration - this test only bounds its deviation (relative to the feature
    magnitude: bf16 has 8 mantissa bits, ~4e-3 per operand) and checks that it is never the default."""
    cfg = _small_cfg(pkg)
    assert cfg.MODEL.ROI_BOX_HEAD.RES5_DTYPE != "bf16"
    cfg.MODEL.ROI_BOX_HEAD.RES5_DTYPE = "bf16"
    heads, params, h = _make_heads(pkg, oracle, cfg, 80, 5)
    rng = np.random.default_rng(5)
    if many:
        feat = rng.standard_normal((2, 128, 20, 30)).astype(np.float32)
        boxes = [oracle.synth_boxes(rng, 300, 480.0, 320.0), oracle.synth_boxes(rng, 260, 480.0, 320.0)]
    else:
        feat = rng.standard_normal((2, 128, 50, 84)).astype(np.float32)
        boxes = [oracle.synth_boxes(rng, 40), oracle.synth_boxes(rng, 33)]
    want = oracle.roi_head_forward(feat, boxes, params, h)
    from locov_amd.structures import Boxes
    with torch.no_grad():
        bf = heads._shared_roi_transform([dev(feat)], [Boxes(torch.from_numpy(b).cuda()) for b in boxes])
        scores, deltas = heads.box_predictor(heads._pooled_mean(bf))
    rel = np.abs(bf.cpu().numpy() - want["res5"]).max() / np.abs(want["res5"]).max()
    assert 0 < rel <= 2e-2, rel                               # > 0: the bf16 path really ran
    srel = np.abs(scores.cpu().numpy() - want["scores"]).max() / np.abs(want["scores"]).max()
    assert srel <= 2e-2, srel


_FULL = {}


def _full_size_oracle(oracle):
    """One 1333x800 image x 1000 proposals through the oracle's ROIAlign + Res5 + mean (a few seconds on the GPU box's
    host cores; computed once per session), plus the inputs."""
    if not _FULL:
        rng = np.random.default_rng(2026)
        feat = rng.standard_normal((1, 1024, 50, 84)).astype(np.float32)
        boxes = [oracle.synth_boxes(rng, 1000)]
        params = oracle.make_res5_params(2026)
        pooled = oracle.roi_pooler([feat], boxes, 14, (1.0 / 16,), 0)
        bf = oracle.spatial_mean(oracle.res5_stage(pooled, params).numpy())
        _FULL.update(feat=feat, boxes=boxes, params=params, box_features=bf)
    return _FULL


@pytest.mark.parametrize("dim", [768, 1024])
@pytest.mark.parametrize("res5_dtype", ["f16x2", "fp32"])
def test_full_size_logits_vs_oracle(pkg, oracle, dim, res5_dtype):
    """BASELINE.json configs 2/3 at their full size against the oracle: 1 x 1000 proposals of a 1333x800 image, Res5
    1024 -> 2048, 1203-class (LVIS-size) bank, D = 768 (reference) and 1024 (north_star), through
    EmbeddingProposalsRes5ROIHeads.inference_detection's pre-NMS path (_shared_roi_transform -> mean -> box_predictor),
    both Res5 arithmetics, fp32 and bf16 similarity GEMM.  Gates: fp32 logits within 1e-4 of the oracle (north_star); bf16
    similarity within 1e-4 of an fp64 product of the SAME bf16-rounded operands."""
    from locov_amd.structures import Boxes, ShapeSpec
    full = _full_size_oracle(oracle)
    cfg = pkg.config.get_cfg()
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    cfg.MODEL.ROI_BOX_HEAD.EMB_DIM = dim
    cfg.MODEL.ROI_BOX_HEAD.RES5_DTYPE = res5_dtype
    cfg.MODEL.ROI_HEADS.NAME = "EmbeddingProposalsRes5ROIHeads"
    heads = pkg.build_roi_heads(cfg, {"res4": ShapeSpec(channels=1024, stride=16)})
    heads.res5.load_state_dict(full["params"])
    h = oracle.synth_head(np.random.default_rng(dim), 2048, dim, 1203)
    bp = heads.box_predictor
    with torch.no_grad():
        bp.emb_pred.weight.copy_(torch.from_numpy(h["emb_w"]))
        bp.emb_pred.bias.copy_(torch.from_numpy(h["emb_b"]))
        bp.bbox_pred.weight.copy_(torch.from_numpy(h["bbox_w"]))
        bp.bbox_pred.bias.copy_(torch.from_numpy(h["bbox_b"]))
    heads = heads.cuda().eval()
    bp.set_class_embeddings(h["cls_w"])
    heads.num_classes = bp.num_classes
    want_scores, want_deltas, want_emb = oracle.box_predictor_forward(full["box_features"], h["emb_w"], h["emb_b"], h["bbox_w"],
                                                                      h["bbox_b"], h["cls_w"])
    boxes = [Boxes(torch.from_numpy(full["boxes"][0]).cuda())]
    assert heads.res5.map_path_pays(1000, 50 * 84)
    with torch.no_grad():
        bf = heads._shared_roi_transform([dev(full["feat"])], boxes, pooled=True)       # roi_emb_heads.py:355-356
        scores, deltas = bp(bf)                                                          # :357
    assert tuple(scores.shape) == (1000, 1204)
    ferr = np.abs(bf.cpu().numpy() - full["box_features"]).max() / np.abs(full["box_features"]).max()
    assert ferr <= 2e-5, ferr
    err = np.abs(scores.cpu().numpy() - want_scores).max()
    assert err <= 1e-4, err                                                              # north_star gate
    np.testing.assert_allclose(deltas.cpu().numpy(), want_deltas, atol=1e-5)
    assert np.all(scores.cpu().numpy()[:, -1] == 0)
    # config 3: bf16 MFMA similarity GEMM on the same region embeddings
    bp.sim_gemm_dtype = "bf16"
    with torch.no_grad():
        scores16, _ = bp(bf)
    bp.sim_gemm_dtype = "fp32"
    # the bf16-rounded operands the device multiplied: its own fp32 region embedding (same GEMM kernel, same bits) and bank
    emb_dev = bp.region_embedding(bf)          # emb_pred in the predictor's own inference arithmetic (split operands under RES5_DTYPE "f16x2")
    assert np.abs(emb_dev.cpu().numpy() - want_emb).max() <= 2e-5 * np.abs(want_emb).max()
    emb16 = emb_dev.cpu().to(torch.bfloat16).double()
    bank16 = torch.from_numpy(h["cls_w"]).to(torch.bfloat16).double()
    want16 = (emb16 @ bank16.t()).numpy()
    err16 = np.abs(scores16.cpu().numpy() - want16).max()
    assert err16 <= 1e-4, err16                                                          # SURVEY.md 8d bf16 gate
    dev16 = np.abs(scores16.cpu().numpy() - want_scores).max()
    assert dev16 <= 5e-2, dev16                                                          # reported deviation from pure fp32


def test_bench_kernel_mix_vs_oracle(pkg, oracle):
    """the kernels the bench line is timed on, against the oracle.  At 1 x 1000 proposals
    (test_full_size_logits_vs_oracle) conv1 and the Winograd-domain GEMMs stay below the 1 024-tile threshold and run the 128x128
    kernel; here 4 x 1000 proposals of 1333x800 images, 1203-class bank, default arithmetic, put every launch kind of
    gemm_split_big_kernel (plain / batched / mean-fused / with the Winograd input transform: timing classes 9 / 10 / 11 / 8,
    include/locov_hip.h) on the path -- asserted -- and the logits must still be within north_star's 1e-4 of
    oracle.roi_head_forward's pieces (ROIAlign 14x14 -> Res5 -> mean -> FCs -> similarity; ~15 s of host time)."""
    import ctypes
    from locov_amd import _lib
    from locov_amd.structures import Boxes, ShapeSpec
    n_img, R = 4, 1000
    rng = np.random.default_rng(404)
    feat = rng.standard_normal((n_img, 1024, 50, 84)).astype(np.float32)
    boxes = [oracle.synth_boxes(rng, R) for _ in range(n_img)]
    params = oracle.make_res5_params(404)
    bfs = []
    for i in range(n_img):                                   # one image at a time: the [1000,1024,14,14] intermediate is 0.8 GB
        pooled = oracle.roi_pooler([feat[i:i + 1]], [boxes[i]], 14, (1.0 / 16,), 0)
        bfs.append(oracle.spatial_mean(oracle.res5_stage(pooled, params).numpy()))
        del pooled
    box_features = np.concatenate(bfs)
    cfg = pkg.config.get_cfg()
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    cfg.MODEL.ROI_HEADS.NAME = "EmbeddingProposalsRes5ROIHeads"
    heads = pkg.build_roi_heads(cfg, {"res4": ShapeSpec(channels=1024, stride=16)})
    assert heads.res5_dtype == "f16x2" and heads.res5_backend == "hip"          # the bench's default arithmetic
    heads.res5.load_state_dict(params)
    h = oracle.synth_head(np.random.default_rng(405), 2048, 768, 1203)
    bp = heads.box_predictor
    with torch.no_grad():
        bp.emb_pred.weight.copy_(torch.from_numpy(h["emb_w"]))
        bp.emb_pred.bias.copy_(torch.from_numpy(h["emb_b"]))
        bp.bbox_pred.weight.copy_(torch.from_numpy(h["bbox_w"]))
        bp.bbox_pred.bias.copy_(torch.from_numpy(h["bbox_b"]))
    heads = heads.cuda().eval()
    bp.set_class_embeddings(h["cls_w"])
    heads.num_classes = bp.num_classes
    want_scores, want_deltas, _ = oracle.box_predictor_forward(box_features, h["emb_w"], h["emb_b"], h["bbox_w"], h["bbox_b"], h["cls_w"])
    lib = _lib.load()
    featd = dev(feat)
    bx = [Boxes(torch.from_numpy(b).cuda()) for b in boxes]
    lib.locov_gemm_timing_enable(1)
    try:
        with torch.no_grad():
            bf = heads._shared_roi_transform([featd], bx, pooled=True)          # roi_emb_heads.py:355-356, as bench.py's step_s2
            scores, deltas = bp(bf)                                             # :357
        torch.cuda.synchronize()
        launches = {}
        for cls in (5, 8, 9, 10, 11, 0):
            n, ms, fl = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            _lib.check(lib.locov_gemm_timing_read(cls, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)))
            launches[cls] = n.value
    finally:
        lib.locov_gemm_timing_enable(0)
    # the bench's mix: 2 x conv1 + input transform (8), conv3 of blocks 0-1 (9), 3 Winograd-domain batched launches (10), the
    # mean-fused conv3 (11) on the 256x256 tile; map GEMM + the two predictor GEMMs on the 128x128 one (5); nothing on the f32 MFMA
    # but bbox_pred (N = 4)
    assert launches[8] == 2 and launches[9] == 2 and launches[10] == 3 and launches[11] == 1, launches
    assert launches[5] == 3, launches
    assert tuple(scores.shape) == (n_img * R, 1204)
    ferr = np.abs(bf.cpu().numpy() - box_features).max() / np.abs(box_features).max()
    assert ferr <= 2e-5, ferr
    err = np.abs(scores.cpu().numpy() - want_scores).max()
    assert err <= 1e-4, err                                                     # north_star gate
    np.testing.assert_allclose(deltas.cpu().numpy(), want_deltas, atol=1e-5)
    assert np.all(scores.cpu().numpy()[:, -1] == 0)


def test_full_size_head_properties(pkg, oracle):
    """BASELINE.json's full size (1333x800 map, 4 x 1000 proposals, Res5 1024 -> 2048, 1203-class bank): the
    size-independent properties of the path (the oracle comparison at 1 x 1000 is test_full_size_logits_vs_oracle):
    image sharding (what bench.py --gpus N does) and proposal permutation leave every row bit-identical, the three
    Res5 forms agree within the logits gate, the background column is exactly zero."""
    cfg = pkg.config.get_cfg()
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    cfg.MODEL.ROI_HEADS.NAME = "EmbeddingProposalsRes5ROIHeads"
    heads, params, h = _make_heads(pkg, oracle, cfg, 1203, 5)
    rng = np.random.default_rng(77)
    feat = dev(rng.standard_normal((4, 1024, 50, 84)).astype(np.float32))
    props, boxes = _proposals(pkg, oracle, rng, 4, 1000)

    def logits(f, pr):
        with torch.no_grad():
            bf = heads._shared_roi_transform([f], [p.proposal_boxes for p in pr])
            return heads.box_predictor(heads._pooled_mean(bf))[0]

    assert heads.res5.map_path_pays(4000, 4 * 50 * 84)
    full = logits(feat, props)
    assert tuple(full.shape) == (4000, 1204) and torch.isfinite(full).all() and torch.all(full[:, -1] == 0)
    # sharding by image: ranks see disjoint image subsets, results must not depend on the split
    a, b = logits(feat[:2], props[:2]), logits(feat[2:], props[2:])
    assert torch.equal(torch.cat([a, b]), full)
    # permuting the proposals of an image permutes its rows
    perm = torch.from_numpy(rng.permutation(1000)).cuda()
    from locov_amd.structures import Boxes, Instances
    p0 = Instances((800, 1333))
    p0.proposal_boxes = Boxes(props[0].proposal_boxes.tensor[perm])
    p0.objectness_logits = props[0].objectness_logits[perm]
    assert torch.equal(logits(feat[:1], [p0]), full[:1000][perm])
    # the pooled-rows form (K-concatenated block 0) and the direct 3x3 form agree within the gate
    heads.res5.map_path_pays = lambda *a, **k: False
    pooled = logits(feat, props)
    heads.res5_conv3x3 = "direct"
    direct = logits(feat, props)
    assert (pooled - full).abs().max().item() <= 1e-4 and (direct - full).abs().max().item() <= 1e-4


def test_training_forward_contract(pkg, oracle):
    """EmbeddingProposalsRes5ROIHeads.forward with targets: 4-tuple, sampled proposals with
    gt_classes / fg_proposal, losses with loss_cls weight 0 under DETACH_CLASS_PREDICTOR
    (roi_emb_heads.py:311-349, box_emb_head.py:147-149)."""
    from locov_amd.structures import Boxes, Instances
    cfg = _small_cfg(pkg)
    cfg.MODEL.ROI_HEADS.DETACH_CLASS_PREDICTOR = True
    cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 32
    cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION = 1.0
    heads, params, h = _make_heads(pkg, oracle, cfg, 80, 9)
    heads.train()
    rng = np.random.default_rng(23)
    feat = dev(rng.standard_normal((2, 128, 50, 84)).astype(np.float32)).requires_grad_(True)
    props, boxes = _proposals(pkg, oracle, rng, 2, 50)
    targets = []
    for i in range(2):
        t = Instances((800, 1333))
        t.gt_boxes = Boxes(torch.from_numpy(boxes[i][:3] + 2.0).cuda())
        t.gt_classes = torch.tensor([1, 1, 1], device="cuda")       # OLN pseudo-GT (coco_mappers.py:88-106)
        targets.append(t)
    grid, box_feats, sampled, losses = heads(None, {"res4": feat}, props, targets)
    assert tuple(grid.shape) == (2, 256, 25, 42)
    assert len(box_feats) == 2 and all(b.shape[1] == 256 for b in box_feats)
    assert all(s.has("gt_classes") and s.has("fg_proposal") and s.has("gt_boxes") for s in sampled)
    assert [len(s) for s in sampled] == [b.shape[0] for b in box_feats]
    assert set(losses) == {"loss_cls", "loss_box_reg"} and float(losses["loss_cls"]) == 0.0
    assert torch.isfinite(losses["loss_box_reg"])
    (losses["loss_box_reg"] + sum(b.sum() for b in box_feats) * 1e-3).backward()
    assert feat.grad is not None and torch.isfinite(feat.grad).all() and feat.grad.abs().sum() > 0
    assert heads.box_predictor.bbox_pred.weight.grad is not None


def _timing_launches(lib, classes):
    import ctypes
    from locov_amd import _lib
    out = {}
    for cls in classes:
        n, ms, fl = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        _lib.check(lib.locov_gemm_timing_read(cls, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)))
        out[cls] = n.value
    return out


def test_size_regime_past_32_bit_offsets(pkg, oracle, monkeypatch):
    """no test ran above 8 000 proposals, and byte offsets pass 2^32 at 10 700 (a [49 R, 2048] fp32
    tensor).  12 x 1000 proposals of 1333x800 images, 1203-class bank, default arithmetic (24 GB of intermediates):
      * the 12 000-proposal call is bit-identical, row for row, to an 8 000- and a 4 000-proposal call (each below 2^32), with
        the mean-fused last convolution ON the path (timing class 11 -- it used to be switched off silently past 10 700
        proposals; the 256x256 kernel addresses from 64-bit tile bases).  (The fused mean adds a proposal's 49 rows in two
        pieces cut where a 64-row chunk of the launch ends, so its LAST BITS depend on the proposal's index modulo 64:
        a split at a multiple of 64 proposals is bit-identical, any other split -- 6 000 + 6 000 -- agrees to fp32
        re-association, asserted at 2e-6);
      * the un-pooled form (plain conv3 on the 4.8 GB tensor, timing class 9 three times, + the stand-alone spatial mean) is
        bit-identical to two 6 000-proposal calls: the sharding identity of test_full_size_head_properties;
      * its first 1 000 rows (image 0 = the inputs of test_full_size_logits_vs_oracle) are within north_star's 1e-4 of the oracle;
      * with the 256x256 tile forbidden (LOCOV_SPLIT_BIG=0: every GEMM of the step on the 128x128 kernels, whose mean-fused
        form DOES keep 32-bit residual offsets) the stage says so once (RuntimeWarning), runs the unfused convolution +
        spatial mean over the 4.8 GB tensor, and agrees within the gate."""
    import warnings
    from locov_amd import _lib
    from locov_amd.structures import Boxes, ShapeSpec
    full = _full_size_oracle(oracle)
    n_img, R = 12, 1000
    rng = np.random.default_rng(1212)
    feat = np.concatenate([full["feat"], rng.standard_normal((n_img - 1, 1024, 50, 84)).astype(np.float32)])
    boxes = [full["boxes"][0]] + [oracle.synth_boxes(rng, R) for _ in range(n_img - 1)]
    cfg = pkg.config.get_cfg()
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    cfg.MODEL.ROI_HEADS.NAME = "EmbeddingProposalsRes5ROIHeads"
    heads = pkg.build_roi_heads(cfg, {"res4": ShapeSpec(channels=1024, stride=16)})
    assert heads.res5_dtype == "f16x2"
    heads.res5.load_state_dict(full["params"])
    h = oracle.synth_head(np.random.default_rng(768), 2048, 768, 1203)
    bp = heads.box_predictor
    with torch.no_grad():
        bp.emb_pred.weight.copy_(torch.from_numpy(h["emb_w"]))
        bp.emb_pred.bias.copy_(torch.from_numpy(h["emb_b"]))
        bp.bbox_pred.weight.copy_(torch.from_numpy(h["bbox_w"]))
        bp.bbox_pred.bias.copy_(torch.from_numpy(h["bbox_b"]))
    heads = heads.cuda().eval()
    bp.set_class_embeddings(h["cls_w"])
    heads.num_classes = bp.num_classes
    featd = dev(feat)
    bx = [Boxes(torch.from_numpy(b).cuda()) for b in boxes]
    lib = _lib.load()

    def run(lo, hi):
        with torch.no_grad():
            bf = heads._shared_roi_transform([featd[lo:hi]], bx[lo:hi], pooled=True)
            return bf, bp(bf)[0]

    assert 49 * n_img * R * 2048 * 4 > 2 ** 32
    lib.locov_gemm_timing_enable(1)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("error", RuntimeWarning)                       # nothing falls back, nothing leaves the range
            bf, scores = run(0, n_img)
        torch.cuda.synchronize()
        launches = _timing_launches(lib, (8, 9, 10, 11))
    finally:
        lib.locov_gemm_timing_enable(0)
    assert launches[11] == 1 and launches[9] == 2 and launches[8] == 2 and launches[10] == 3, launches
    assert tuple(scores.shape) == (n_img * R, 1204) and bool(torch.isfinite(scores).all())
    parts = [run(0, 8), run(8, 12)]                                              # 8 000 = 125 x 64 proposals
    assert torch.equal(torch.cat([x[0] for x in parts]), bf)
    assert torch.equal(torch.cat([x[1] for x in parts]), scores)
    parts = [run(0, 6), run(6, 12)]
    assert float((torch.cat([x[0] for x in parts]) - bf).abs().max() / bf.abs().max()) <= 2e-6
    del parts

    def run_unpooled(lo, hi):
        with torch.no_grad():
            return heads._pooled_mean(heads._shared_roi_transform([featd[lo:hi]], bx[lo:hi]))

    lib.locov_gemm_timing_enable(1)
    try:
        bfu = run_unpooled(0, n_img)
        torch.cuda.synchronize()
        launches = _timing_launches(lib, (9, 11))
    finally:
        lib.locov_gemm_timing_enable(0)
    assert launches[9] == 3 and launches[11] == 0, launches
    assert torch.equal(torch.cat([run_unpooled(0, 6), run_unpooled(6, 12)]), bfu)
    assert float((bfu - bf).abs().max() / bf.abs().max()) <= 2e-6
    del bfu
    want_scores, _, _ = oracle.box_predictor_forward(full["box_features"], h["emb_w"], h["emb_b"], h["bbox_w"], h["bbox_b"], h["cls_w"])
    ferr = np.abs(bf[:R].cpu().numpy() - full["box_features"]).max() / np.abs(full["box_features"]).max()
    assert ferr <= 2e-5, ferr
    assert np.abs(scores[:R].cpu().numpy() - want_scores).max() <= 1e-4
    # the 128x128 kernels on the same 4.8 GB operands
    monkeypatch.setenv("LOCOV_SPLIT_BIG", "0")
    heads.res5.__dict__.pop("_warned", None)
    with pytest.warns(RuntimeWarning, match="NOT fused"):
        bf128, scores128 = run(0, n_img)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)                           # ... once
        run(0, 1)
    monkeypatch.delenv("LOCOV_SPLIT_BIG")
    rel = float((bf128 - bf).abs().max() / bf.abs().max())
    assert rel <= 1e-5, rel
    assert float((scores128 - scores).abs().max()) <= 1e-4
