"""Split-operand fp32 GEMM (locov_gemm_nt_f32_split / _batched_ / locov_winograd_conv3x3_f32_split; csrc/gemm_split.hip):
fp32 in, fp32 out, products formed on the f16 matrix pipe from (hi, lo) fp16 pairs.  It stands in for the same
reference convolutions as the fp32-MFMA GEMM (Res5, roi_emb_heads.py:217-245), so the bar is the fp32 one: its error
against an fp64 product must be of the order of the fp32-MFMA kernel's on the same inputs, over the whole range of
magnitudes the operand scales are meant to cover, and the head's logits must stay inside the 1e-4 gate."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device")
    from locov_amd import _lib, ops
    _lib.load()
    return ops


def _rel(y, ref):
    return float((y.double() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("M,N,K", [(1, 4, 32), (127, 128, 64), (128, 132, 96), (1000, 512, 2048), (777, 2048, 512),
                                   (333, 2560, 1024), (4097, 388, 1536)])
@pytest.mark.parametrize("mag", [1.0, 0.05, 10.0])
def test_split_gemm_is_as_accurate_as_the_fp32_mfma_gemm(ops, M, N, K, mag):
    g = torch.Generator().manual_seed(M * 7 + N)
    x = (torch.randn(M, K, generator=g) * mag).relu_()
    x[::7, ::5] *= 1e-4                                       # small entries next to large ones
    w = torch.randn(N, K, generator=g) * 0.02
    ref = x.double() @ w.double().t()
    xd, wd = x.cuda(), w.cuda()
    e32 = _rel(ops.linear(xd, wd).cpu(), ref)
    es = _rel(ops.linear_split(xd, ops.split_pack(wd)).cpu(), ref)       # default x_scale = 16: |x| < 4094
    # 22-bit operands: ~2.4e-7 per operand on top of the accumulation error both kernels share
    assert es <= 3e-6, es
    assert es <= 1.5 * e32 + 6e-7, (es, e32)


def test_split_gemm_operand_scale_covers_other_magnitudes(ops):
    """x_scale places the activations in fp16's range: 2^-6-scaled for values up to 4e6, 2^12-scaled for 1e-5-sized
    ones -- same accuracy as at unit scale."""
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(256, 512, generator=g) * 0.02).cuda()
    for mag, xs in ((1e-5, 2.0 ** 22), (3e4, 2.0 ** -6)):
        x = (torch.randn(500, 512, generator=g) * mag).relu_().cuda()
        ref = x.double() @ w.double().t()
        es = _rel(ops.linear_split(x, ops.split_pack(w), x_scale=xs), ref)
        assert es <= 2e-6, (mag, es)


def test_split_pack_layout_and_scale(ops):
    """scale * w == hi + lo to 22 bits; per group of 8 columns the packed row holds 8 hi halves, then 8 lo halves."""
    g = torch.Generator().manual_seed(3)
    w = torch.randn(5, 64, generator=g) * 0.03
    sw = ops.split_pack(w.cuda())
    assert sw.data.shape == w.shape and sw.data.dtype == torch.float32
    amax = float(w.abs().max()) * sw.scale
    assert 2.0 ** 12 <= amax < 2.0 ** 13 and np.log2(sw.scale) == int(np.log2(sw.scale))
    halves = sw.data.cpu().view(torch.float16).view(5, 8, 2, 8).float()      # [row, group, hi/lo, 8]
    back = (halves[:, :, 0] + halves[:, :, 1]).reshape(5, 64) / sw.scale
    assert (back - w).abs().max().item() <= 2.0 ** -21 * float(w.abs().max())
    np.testing.assert_array_equal(halves[:, :, 0].reshape(5, 64).numpy(), (w * sw.scale).half().float().numpy())


def test_split_gemm_epilogue_and_strided_rows(ops):
    """scale / shift / residual / ReLU, x as a column block of a wider matrix, ragged M and N tiles."""
    g = torch.Generator().manual_seed(11)
    M, N, K = 901, 516, 160
    wide = torch.randn(M, K + 64, generator=g).cuda()
    x = wide[:, 32:32 + K]
    w = (torch.randn(N, K, generator=g) * 0.05).cuda()
    sc, sh, res = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    ref = torch.relu((x.double() @ w.double().t()) * sc.double() + sh.double() + res.double())
    got = ops.linear_split(x, ops.split_pack(w), sh, scale=sc, residual=res, relu=True)
    assert (got.double() - ref).abs().max().item() <= 5e-6
    plain = ops.linear_split(x, ops.split_pack(w))
    assert (plain.double() - x.double() @ w.double().t()).abs().max().item() <= 5e-6


def test_split_gemm_rejects_what_it_cannot_run(ops):
    from locov_amd._lib import LocovError
    x = torch.randn(8, 48).cuda()
    with pytest.raises(ValueError):
        ops.split_pack(x)                                     # K % 32
    w = ops.split_pack(torch.randn(6, 64).cuda())
    with pytest.raises(ValueError):
        ops.linear_split(torch.randn(8, 64).cuda(), w)        # N % 4
    with pytest.raises((LocovError, TypeError)):
        ops.linear_split(torch.randn(8, 64), w)               # host tensor: no CPU path


@pytest.mark.parametrize("B,M,N,K", [(3, 70, 48, 64), (121, 130, 128, 32), (5, 700, 512, 512)])
def test_split_batched_gemm(ops, B, M, N, K):
    g = torch.Generator().manual_seed(B * 1000 + M)
    x = torch.randn(B, M, K, generator=g) * 10
    w = torch.randn(B, N, K, generator=g) * 0.05
    ref = torch.bmm(x.double(), w.double().transpose(1, 2))
    got = ops.gemm_nt_batched_split(x.cuda(), ops.split_pack(w.cuda())).cpu()
    e32 = _rel(ops.gemm_nt_batched(x.cuda(), w.cuda()).cpu(), ref)
    es = _rel(got, ref)
    assert es <= 3e-6 and es <= 1.5 * e32 + 2e-7, (es, e32)


@pytest.mark.parametrize("R,Cin,N", [(1, 32, 4), (37, 64, 48), (129, 512, 512)])
def test_split_winograd_conv_vs_direct(ops, R, Cin, N):
    g = torch.Generator().manual_seed(R)
    x = torch.randn(R, Cin, 7, 7, generator=g).relu_()
    w = torch.randn(N, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    sc, sh = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1
    ref = torch.relu(F.conv2d(x.double(), w.double(), padding=1) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    rows = x.permute(2, 3, 0, 1).reshape(49 * R, Cin).contiguous().cuda()
    U = ops.winograd_pack_weight(w.cuda())
    y32 = ops.winograd_conv3x3(rows, U, scale=sc.cuda(), shift=sh.cuda(), relu=True)
    ys = ops.winograd_conv3x3(rows, ops.split_pack(U), scale=sc.cuda(), shift=sh.cuda(), relu=True)
    unrows = lambda y: y.reshape(7, 7, R, -1).permute(2, 3, 0, 1).cpu()
    e32, es = _rel(unrows(y32), ref), _rel(unrows(ys), ref)
    assert es <= 2e-5 and es <= 1.5 * e32 + 1e-6, (es, e32)


@pytest.mark.parametrize("dtype", ["f16x2", "fp32"])
@pytest.mark.parametrize("many", [False, True])
def test_heads_pass_the_logit_gate_with_either_res5_arithmetic(many, dtype):
    """MODEL.ROI_BOX_HEAD.RES5_DTYPE = "f16x2" (default) and "fp32": the whole head against the CPU oracle, the same
    gates for both (logits 1e-4, box deltas 1e-5), on the pooled-rows path and on the map path."""
    import locov_amd as pkg
    from oracle import lsm_oracle as oracle
    import test_gpu_roi_heads as T
    oracle.build()
    cfg = T._small_cfg(pkg)
    assert cfg.MODEL.ROI_BOX_HEAD.RES5_DTYPE == "f16x2"
    cfg.MODEL.ROI_BOX_HEAD.RES5_DTYPE = dtype
    heads, params, h = T._make_heads(pkg, oracle, cfg, 80, 23)
    rng = np.random.default_rng(23)
    if many:
        feat = rng.standard_normal((2, 128, 20, 30)).astype(np.float32)
        boxes = [oracle.synth_boxes(rng, 300, 480.0, 320.0), oracle.synth_boxes(rng, 260, 480.0, 320.0)]
    else:
        feat = rng.standard_normal((2, 128, 50, 84)).astype(np.float32)
        boxes = [oracle.synth_boxes(rng, 40), oracle.synth_boxes(rng, 33)]
    want = oracle.roi_head_forward(feat, boxes, params, h)
    from locov_amd.structures import Boxes
    with torch.no_grad():
        bf = heads._shared_roi_transform([T.dev(feat)], [Boxes(torch.from_numpy(b).cuda()) for b in boxes])
        scores, deltas = heads.box_predictor(heads._pooled_mean(bf))
    assert np.abs(bf.cpu().numpy() - want["res5"]).max() <= 2e-5 * np.abs(want["res5"]).max()
    assert np.abs(scores.cpu().numpy() - want["scores"]).max() <= 1e-4
    np.testing.assert_allclose(deltas.cpu().numpy(), want["deltas"], atol=1e-5)


@pytest.mark.parametrize("dtype", ["f16x2", "fp32"])
def test_reference_config_heads_with_either_res5_arithmetic(dtype):
    """configs/coco_lsm.yaml shapes (Res5 1024 -> 2048, D = 768, 80-class bank): logits within 1e-4 of the oracle."""
    import locov_amd as pkg
    from oracle import lsm_oracle as oracle
    import test_gpu_roi_heads as T
    oracle.build()
    cfg = pkg.config.get_cfg()
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    cfg.MODEL.ROI_HEADS.NAME = "EmbeddingProposalsRes5ROIHeads"
    cfg.MODEL.ROI_BOX_HEAD.RES5_DTYPE = dtype
    heads, params, h = T._make_heads(pkg, oracle, cfg, 80, 1992)
    rng = np.random.default_rng(1992)
    feat = rng.standard_normal((2, 1024, 50, 84)).astype(np.float32)
    props, boxes = T._proposals(pkg, oracle, rng, 2, 40)
    want = oracle.roi_head_forward(feat, boxes, params, h)
    with torch.no_grad():
        bf = heads._shared_roi_transform([T.dev(feat)], [p.proposal_boxes for p in props])
        scores, deltas = heads.box_predictor(heads._pooled_mean(bf))
    err = np.abs(scores.cpu().numpy() - want["scores"]).max()
    assert err <= 1e-4, err
    np.testing.assert_allclose(deltas.cpu().numpy(), want["deltas"], atol=1e-5)


@pytest.mark.parametrize("fscale", [30.0, 0.03])
def test_split_res5_tracks_the_f32_mfma_path_across_input_magnitudes(fscale):
    """Heavy-tailed post-ReLU features scaled by 30 (Res5 activations up to ~1.4e3) and by 0.03: the split-operand head's
    error against the oracle stays at the f32-MFMA head's (relative to the logit range, which scales with the input)."""
    import locov_amd as pkg
    from oracle import lsm_oracle as oracle
    import test_gpu_roi_heads as T
    oracle.build()
    errs = {}
    for dtype in ("fp32", "f16x2"):
        cfg = T._small_cfg(pkg)
        cfg.MODEL.ROI_BOX_HEAD.RES5_DTYPE = dtype
        heads, params, h = T._make_heads(pkg, oracle, cfg, 80, 1992)
        rng = np.random.default_rng(1992)
        feat = (np.maximum(rng.standard_normal((2, 128, 50, 84)), 0) * fscale).astype(np.float32)
        feat[:, :, ::9, ::7] *= 8.0
        props, boxes = T._proposals(pkg, oracle, rng, 2, 60)
        want = oracle.roi_head_forward(feat, boxes, params, h)
        with torch.no_grad():
            bf = heads._shared_roi_transform([T.dev(feat)], [p.proposal_boxes for p in props])
            scores, _ = heads.box_predictor(heads._pooled_mean(bf))
        errs[dtype] = np.abs(scores.cpu().numpy() - want["scores"]).max() / np.abs(want["scores"]).max()
    assert errs["f16x2"] <= 5e-6 and errs["f16x2"] <= 2 * errs["fp32"] + 5e-7, errs


@pytest.mark.parametrize("R,N,K", [(1, 4, 32), (5, 128, 64), (300, 516, 160), (131, 2048, 512)])
def test_segmean_matches_the_unfused_gemm_plus_mean(ops, R, N, K):
    """locov_gemm_nt_f32_split_segmean: ROI-major x rows, POSITION-major residual -> mean over the 49 positions of the
    finished values; against the unfused split GEMM + mean and an fp64 evaluation; bit-identical from run to run."""
    seg, M = 49, 49 * R
    g = torch.Generator().manual_seed(R + N)
    x = torch.randn(M, K, generator=g).relu_().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).cuda()
    res_pm = torch.randn(M, N, generator=g).cuda()
    sc, sh = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda()
    res_rm = res_pm.view(seg, R, N).permute(1, 0, 2).reshape(M, N).contiguous()
    ws = ops.split_pack(w)
    want = ops.linear_split(x, ws, sh, scale=sc, residual=res_rm, relu=True).view(R, seg, N).double().mean(dim=1)
    ref = torch.relu((x.double() @ w.double().t()) * sc.double() + sh.double() + res_rm.double()).view(R, seg, N).mean(dim=1)
    got = ops.linear_split_segmean(x, ws, sh, res_pm, seg, scale=sc, relu=True)
    assert tuple(got.shape) == (R, N)
    assert (got.double() - want).abs().max().item() <= 2e-6
    assert (got.double() - ref).abs().max().item() <= 3e-6
    assert torch.equal(got, ops.linear_split_segmean(x, ws, sh, res_pm, seg, scale=sc, relu=True))


def test_winograd_roi_major_output(ops):
    g = torch.Generator().manual_seed(9)
    R, Cin, N = 37, 64, 48
    rows = torch.randn(49 * R, Cin, generator=g).cuda()
    U = ops.winograd_pack_weight((torch.randn(N, Cin, 3, 3, generator=g) * 0.1).cuda())
    for u in (U, ops.split_pack(U)):
        pm = ops.winograd_conv3x3(rows, u, relu=True)
        rm = ops.winograd_conv3x3(rows, u, relu=True, roi_major=True)
        assert torch.equal(rm.view(R, 49, N), pm.view(49, R, N).permute(1, 0, 2))


@pytest.mark.parametrize("many", [False, True])
def test_pooled_stage_output_equals_the_mean_of_the_unpooled_one(many):
    """Res5Stage.forward_rows / forward_from_map(pooled=True) (fused mean in split arithmetic) against the mean of the
    rows the same call returns without `pooled`, and the heads' pooled path against the oracle."""
    import locov_amd as pkg
    from oracle import lsm_oracle as oracle
    import test_gpu_roi_heads as T
    oracle.build()
    cfg = T._small_cfg(pkg)
    heads, params, h = T._make_heads(pkg, oracle, cfg, 80, 41)
    rng = np.random.default_rng(41)
    if many:
        feat = rng.standard_normal((2, 128, 20, 30)).astype(np.float32)
        boxes = [oracle.synth_boxes(rng, 300, 480.0, 320.0), oracle.synth_boxes(rng, 260, 480.0, 320.0)]
    else:
        feat = rng.standard_normal((2, 128, 50, 84)).astype(np.float32)
        boxes = [oracle.synth_boxes(rng, 40), oracle.synth_boxes(rng, 33)]
    want = oracle.roi_head_forward(feat, boxes, params, h)
    from locov_amd.structures import Boxes
    bl = [Boxes(torch.from_numpy(b).cuda()) for b in boxes]
    with torch.no_grad():
        full = heads._shared_roi_transform([T.dev(feat)], bl)
        pooled = heads._shared_roi_transform([T.dev(feat)], bl, pooled=True)
        scores, deltas = heads.box_predictor(pooled)
    assert tuple(pooled.shape) == (full.shape[0], full.shape[1])
    assert (pooled - heads._pooled_mean(full)).abs().max().item() <= 2e-6 * float(full.abs().max())
    np.testing.assert_allclose(pooled.cpu().numpy(), want["box_features"], atol=2e-5 * np.abs(want["res5"]).max(), rtol=1e-4)
    assert np.abs(scores.cpu().numpy() - want["scores"]).max() <= 1e-4
    np.testing.assert_allclose(deltas.cpu().numpy(), want["deltas"], atol=1e-5)


def test_split_gemm_overflow_is_loud(ops):
    """Outside the range the operand scale covers (|x_scale * x| >= 65504) the split GEMM does not return a plausible
    wrong answer: the affected outputs are inf / NaN.  The same data is fine with a smaller x_scale."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(64, 64, generator=g).cuda()
    x[3, 5] = 1.0e5                                            # 16 * 1e5 > 65504
    w = ops.split_pack((torch.randn(32, 64, generator=g) * 0.1).cuda())
    ops.split_overflow_reset(x.device)
    y = ops.linear_split(x, w)                                 # default x_scale = 16
    assert not torch.isfinite(y[3]).all() and torch.isfinite(y[:3]).all() and torch.isfinite(y[4:]).all()
    assert ops.split_overflow_raised(x.device)                 # ... and the launch raised the range-guard word
    assert ops.split_overflow_raised(x.device)                 # (sticky until reset)
    ops.split_overflow_reset(x.device)
    assert torch.isfinite(ops.linear_split(x, w, x_scale=0.25)).all()
    assert not ops.split_overflow_raised(x.device)
    # just inside the range: 16 * 4093 < 65504 -- finite, no flag
    x[3, 5] = 4093.0
    assert torch.isfinite(ops.linear_split(x, w)).all() and not ops.split_overflow_raised(x.device)
    x[3, 5] = -4095.0
    ops.linear_split(x, w)
    assert ops.split_overflow_raised(x.device)
    ops.split_overflow_reset(x.device)
    # the batched (Winograd-domain) and mean-fused forms carry the same guard
    xb = torch.randn(3, 70, 64, generator=g).cuda()
    wb = ops.split_pack((torch.randn(3, 32, 64, generator=g) * 0.1).cuda())
    ops.gemm_nt_batched_split(xb, wb, x_scale=1.0)
    assert not ops.split_overflow_raised(x.device)
    xb[2, 69, 63] = 7.0e4
    ops.gemm_nt_batched_split(xb, wb, x_scale=1.0)
    assert ops.split_overflow_raised(x.device)
    ops.split_overflow_reset(x.device)


def test_heads_repeat_an_out_of_range_call_on_the_f32_mfma():
    """The default arithmetic is safe as a drop-in: a Res5 activation beyond the split arithmetic's range (|x| >= 4094;
    here a FrozenBN with heavy-tailed scales up to 10 on a map with entries up to ~5000, as a trained checkpoint may have)
    does NOT turn into NaN / dropped detections -- the heads read the range-guard word once per call, warn once, and repeat
    that call on the f32 MFMA; the result equals the RES5_DTYPE 'fp32' heads' bit for bit.  In-range calls are not repeated."""
    import warnings
    import locov_amd as pkg
    from oracle import lsm_oracle as oracle
    import test_gpu_roi_heads as T
    oracle.build()
    cfg = T._small_cfg(pkg)
    heads, params, h = T._make_heads(pkg, oracle, cfg, 80, 3)
    cfg32 = T._small_cfg(pkg)
    cfg32.MODEL.ROI_BOX_HEAD.RES5_DTYPE = "fp32"
    heads32, _, _ = T._make_heads(pkg, oracle, cfg32, 80, 3)
    rng = np.random.default_rng(3)
    with torch.no_grad():
        for hd in (heads, heads32):                        # heavy-tailed FrozenBN scale on block 0's conv1
            hd.res5[0].conv1.norm.weight.copy_(torch.from_numpy(np.geomspace(0.5, 10.0, 64).astype(np.float32)))
    feat = rng.standard_normal((2, 128, 50, 84)).astype(np.float32)
    props, boxes = T._proposals(pkg, oracle, rng, 2, 150)
    calls = []
    orig = heads._fused_roi_transform
    heads._fused_roi_transform = lambda f, b, p, dt: (calls.append(dt), orig(f, b, p, dt))[1]

    def run(hd, f):
        # (the predictor is kept on the f32 MFMA for both heads: what is compared here is the guarded Res5 call)
        with torch.no_grad():
            return hd.box_predictor(hd._shared_roi_transform([T.dev(f)], [p.proposal_boxes for p in props], pooled=True), force_fp32=True)[0]

    with warnings.catch_warnings():
        warnings.simplefilter("error")                     # in range: no warning, one pass
        ok = run(heads, feat)
    assert calls == ["f16x2"] and torch.isfinite(ok).all()
    assert (ok - run(heads32, feat)).abs().max() <= 1e-4
    big = feat.copy()
    big[1, :, 20:24, 30:36] *= 1500.0                      # activations of several thousand in one image region
    calls.clear()
    with pytest.warns(RuntimeWarning, match="repeated"):
        got = run(heads, big)
    assert calls == ["f16x2", "fp32"]
    want = run(heads32, big)
    assert torch.isfinite(got).all() and torch.equal(got, want)
    calls.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("error")                     # reported once per module; the retry itself still happens
        got2 = run(heads, big)
    assert calls == ["f16x2", "fp32"] and torch.equal(got2, want)
    # with the check switched off the hazard is visible: the inf / NaN of the affected GEMM rows pass through ReLUs
    # (fmaxf(NaN, 0) = 0), so the logits can even come out FINITE and wrong -- which is why the guard looks at the operands
    # inside the kernel and not at the results
    heads.res5_overflow_check = False
    calls.clear()
    raw = run(heads, big)
    assert calls == ["f16x2"]
    assert (not torch.isfinite(raw).all()) or float((raw - want).abs().max()) > 1e-2


@pytest.mark.parametrize("split", [True, False])
def test_roi_major_stage_equals_the_position_major_one(split):
    """Res5Stage.forward_from_map(roi_major=True): ROIAlign, the Winograd transforms and the mean-fused last convolution
    all in ROI-major row order -- same values as the position-major pipeline, rows permuted; pooled outputs equal."""
    import locov_amd as pkg
    from locov_amd import ops
    from locov_amd.res5 import build_res5_block
    from oracle import lsm_oracle as oracle
    import test_gpu_roi_heads as T
    oracle.build()
    res5, out_ch = build_res5_block(T._small_cfg(pkg))
    res5.load_state_dict(oracle.make_res5_params(13, in_ch=128, mid=64, out_ch=256))
    res5 = res5.cuda().eval()
    rng = np.random.default_rng(3)
    feat = rng.standard_normal((2, 128, 50, 84)).astype(np.float32)
    rois = T.dev(oracle.boxes_to_pooler_format([oracle.synth_boxes(rng, 90), oracle.synth_boxes(rng, 75)]))
    R = rois.shape[0]
    with torch.no_grad():
        nhwc = ops.nchw_to_nhwc(T.dev(feat))
        pm = res5.forward_from_map(nhwc, rois, 14, 1.0 / 16, 0, True, split=split)
        rm = res5.forward_from_map(nhwc, rois, 14, 1.0 / 16, 0, True, split=split, roi_major=True)
        scale = float(pm.abs().max())
        assert (rm.view(R, 49, out_ch) - pm.view(49, R, out_ch).permute(1, 0, 2)).abs().max().item() <= 2e-6 * scale
        pp = res5.forward_from_map(nhwc, rois, 14, 1.0 / 16, 0, True, split=split, pooled=True)
        rp = res5.forward_from_map(nhwc, rois, 14, 1.0 / 16, 0, True, split=split, pooled=True, roi_major=True)
        assert tuple(rp.shape) == (R, out_ch) and (rp - pp).abs().max().item() <= 2e-6 * scale
        assert (rp - rm.view(R, 49, out_ch).mean(dim=1)).abs().max().item() <= 2e-6 * scale


def test_split_gemm_random_shapes(ops):
    """40 random (M, N, K, strided lda, epilogue) combinations against fp64: ragged M / N tiles, one to many K-tiles."""
    rng = np.random.default_rng(2024)
    for _ in range(40):
        M = int(rng.integers(1, 700))
        N = 4 * int(rng.integers(1, 130))
        K = 32 * int(rng.integers(1, 9))
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
        got = ops.linear_split(x, ops.split_pack(w), sh, scale=sc, residual=res, relu=relu)
        err = (got.double() - ref).abs().max().item()
        assert err <= 4e-6 * max(1.0, float(ref.abs().max())), (M, N, K, pad, use_res, use_aff, relu, err)


def test_segmean_random_shapes_both_residual_orders(ops):
    rng = np.random.default_rng(77)
    for _ in range(16):
        R, seg = int(rng.integers(1, 90)), int(rng.choice([49, 49, 43, 64, 100, 128]))
        N, K = 4 * int(rng.integers(1, 80)), 32 * int(rng.integers(1, 6))
        M = R * seg
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        x = torch.randn(M, K, generator=g).relu_().cuda()
        w = (torch.randn(N, K, generator=g) * 0.05).cuda()
        res_rm = torch.randn(M, N, generator=g).cuda()                                  # ROI-major rows q*seg + p
        res_pm = res_rm.view(R, seg, N).permute(1, 0, 2).reshape(M, N).contiguous()     # position-major rows p*R + q
        sh = torch.randn(N, generator=g).cuda()
        ref = torch.relu(x.double() @ w.double().t() + sh.double() + res_rm.double()).view(R, seg, N).mean(dim=1)
        ws = ops.split_pack(w)
        a = ops.linear_split_segmean(x, ws, sh, res_pm, seg)
        b = ops.linear_split_segmean(x, ws, sh, res_rm, seg, residual_roi_major=True)
        for got in (a, b):
            assert (got.double() - ref).abs().max().item() <= 4e-6 * max(1.0, float(ref.abs().max())), (R, seg, N, K)
        assert torch.equal(a, b)
    from locov_amd._lib import LocovError
    with pytest.raises(LocovError):                           # fewer than 43 rows per ROI: more than four ROIs per tile
        ops.linear_split_segmean(torch.randn(42, 32).cuda(), ops.split_pack(torch.randn(8, 32).cuda()), None,
                                 torch.randn(42, 8).cuda(), 7)


@pytest.mark.parametrize("M,N,K", [(300, 64, 64), (1000, 512, 2048), (777, 2048, 512), (129, 136, 96)])
def test_split_layout_output_and_residual(ops, M, N, K):
    """LOCOV_EPI_OUT_SPLIT / LOCOV_EPI_RES_SPLIT: a GEMM writes its finished values in the split layout (the pre-split A and the
    residual of the next block) and reads a residual from it.  Unpacked, the output equals the fp32 output to 2^-22 relative
    (what the consuming GEMM's own split would keep anyway); as a residual it gives the fp32-residual result to the same
    precision; as a pre-split A it gives BIT-identical results to converting the unpacked values in the kernel; values
    outside fp16's range raise the guard."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.relu(torch.randn(M, K, generator=g)).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    sc, sh = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda()
    wp = ops.split_pack(w)
    y32 = ops.linear_split(x, wp, sh, scale=sc, residual=res, relu=True)
    ops.split_overflow_reset(x.device)
    ysp = ops.linear_split(x, wp, sh, scale=sc, residual=res, relu=True, out_split=True)
    assert not ops.split_overflow_raised(x.device)
    got = ops.split_unpack(ysp, 16.0)
    tol = 2.0 ** -21 * y32.abs().clamp_min(2.0 ** -7)                      # 22 bits, absolute floor 2^-25/16 * ... below 2^-7
    assert bool(((got - y32).abs() <= tol).all()), float(((got - y32).abs() / tol).max())
    # the split tensor as the residual of another GEMM == its unpacked values as an fp32 residual (bit for bit: hi + lo is exact)
    if N % 32 == 0:
        w2 = (torch.randn(N, N, generator=g) * 0.05).cuda()
        w2p = ops.split_pack(w2)
        a = torch.relu(torch.randn(M, N, generator=g)).cuda()
        r1 = ops.linear_split(a, w2p, residual=ysp, relu=True, residual_is_split=True)
        r2 = ops.linear_split(a, w2p, residual=got, relu=True)
        assert torch.equal(r1, r2)
        # ... and as the pre-split A operand ~ converting the unpacked values in the kernel: the same hi / lo halves except where
        # hi + lo sits exactly on an fp16 rounding tie (the re-split then picks the other neighbour: same value, another pair)
        z1 = ops.linear_split(ysp, w2p, relu=True, x_is_split=True, x_scale=16.0)
        z2 = ops.linear_split(got, w2p, relu=True, x_scale=16.0)
        assert float((z1 - z2).abs().max()) <= 1e-6 * float(z2.abs().max())
        # mean-fused form with a split residual
        if M % 49 == 0 or True:
            Mr = (M // 49) * 49
            if Mr:
                s1 = ops.linear_split_segmean(a[:Mr], w2p, None, ysp[:Mr], 49, residual_roi_major=True, residual_is_split=True)
                s2 = ops.linear_split_segmean(a[:Mr], w2p, None, got[:Mr].contiguous(), 49, residual_roi_major=True)
                assert torch.equal(s1, s2)
    # range guard on the OUTPUT
    big = x.clone()
    big[3, :] = 4000.0
    ops.split_overflow_reset(x.device)
    ops.linear_split(big, ops.split_pack(torch.ones(N, K).cuda()), out_split=True, x_scale=1.0)   # outputs ~4000*K >> 65504
    assert ops.split_overflow_raised(x.device)
    ops.split_overflow_reset(x.device)


def test_big_tile_kernel_is_bit_identical_to_the_128x128_kernel(ops):
    """gemm_split_big.hip (256 x 256 tile, one 8-wave workgroup per CU; taken by launches with both operands pre-split that are
    large enough) against gemm_split.hip's 128 x 128 kernel on the same pre-split operands: every accumulator sees the same
    products in the same order, so the results are BIT-identical -- ragged M and N tiles, K from 64 up, residual (fp32 and
    split layout), FrozenBN scale / shift, ReLU, split-layout output, and the batched (Winograd-domain) form.
    LOCOV_SPLIT_BIG is read by the launcher at every call: 1 = whenever legal, 0 = never."""
    import os
    rng = np.random.default_rng(808)
    prev = os.environ.get("LOCOV_SPLIT_BIG")
    try:
        for it in range(14):
            M = int(rng.integers(1, 1500))
            N = 8 * int(rng.integers(1, 90))
            K = 64 * int(rng.integers(1, 9))
            g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
            xs = ops.split_pack(torch.relu(torch.randn(M, K, generator=g)).cuda(), 16.0).data
            wp = ops.split_pack((torch.randn(N, K, generator=g) * 0.05).cuda())
            use_res, use_aff, relu, out_split, res_split = (bool(rng.integers(2)) for _ in range(5))
            if use_res and res_split:
                N = 32 * max(1, N // 32)                 # (split_pack of the stand-in residual wants whole 32-column groups)
                wp = ops.split_pack((torch.randn(N, K, generator=g) * 0.05).cuda())
            res = torch.randn(M, N, generator=g).cuda() if use_res else None
            if use_res and res_split:
                res = ops.split_pack(res.abs(), 16.0).data
            sc = (torch.rand(N, generator=g) + 0.5).cuda() if use_aff else None
            sh = torch.randn(N, generator=g).cuda() if use_aff else None
            outs = []
            for big in ("0", "1"):
                os.environ["LOCOV_SPLIT_BIG"] = big
                outs.append(ops.linear_split(xs, wp, sh, scale=sc, residual=res, relu=relu, x_scale=16.0, x_is_split=True,
                                             out_split=out_split, residual_is_split=use_res and res_split))
            assert torch.equal(outs[0], outs[1]), (it, M, N, K, use_res, res_split, use_aff, relu, out_split)
        # the batched launch inside the Winograd-domain convolution
        x = torch.relu(torch.randn(49 * 37, 64, generator=torch.Generator().manual_seed(4))).cuda()
        u = ops.split_pack(ops.winograd_pack_weight((torch.randn(96, 64, 3, 3, generator=torch.Generator().manual_seed(5)) * 0.05).cuda()))
        ys = []
        for big in ("0", "1"):
            os.environ["LOCOV_SPLIT_BIG"] = big
            ys.append(ops.winograd_conv3x3(x, u, relu=True, roi_major=True, in_roi_major=True))
        assert torch.equal(ys[0], ys[1])
    finally:
        if prev is None:
            os.environ.pop("LOCOV_SPLIT_BIG", None)
        else:
            os.environ["LOCOV_SPLIT_BIG"] = prev


def test_predictor_fcs_in_split_arithmetic_match_the_f32_mfma():
    """EmbeddingFastRCNNOutputLayers in inference under RES5_DTYPE "f16x2": emb_pred and cls_score run as split-operand GEMMs
    (fc_dtype "f16x2").  Same logits as the f32-MFMA form (force_fp32=True) to well inside the 1e-4 gate, at the config shape
    (2048 -> 768, 1204-row bank), with and without normalisation; bbox deltas identical (that FC stays on the f32 MFMA); the
    weights are read at call time (re-assigned emb_pred.weight, swapped bank)."""
    import locov_amd as pkg
    from locov_amd.structures import ShapeSpec
    cfg = pkg.config.get_cfg()
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    for norm in (False, True):
        cfg.MODEL.ROI_BOX_HEAD.NORMALIZE_EMB_PRED = norm
        torch.manual_seed(3)
        pred = pkg.roi_heads.build_box_predictor(cfg, 2048).cuda().eval()
        assert pred.fc_dtype == "f16x2"
        g = torch.Generator().manual_seed(11)
        bank = torch.randn(1204, 768, generator=g) * 0.05
        bank[-1] = 0
        pred.set_class_embeddings(bank)
        x = (torch.relu(torch.randn(3000, 2048, generator=g)) * 1.5).cuda()
        with torch.no_grad():
            s_split, d_split = pred(x)
            s_f32, d_f32 = pred(x, force_fp32=True)
        assert torch.equal(d_split, d_f32)
        assert float((s_split - s_f32).abs().max()) <= 2e-5 * max(1.0, float(s_f32.abs().max()))
        assert torch.all(s_split[:, -1] == 0)                              # the zero background row stays exactly 0
        # weights are read at call time
        with torch.no_grad():
            pred.emb_pred.weight = torch.nn.Parameter(pred.emb_pred.weight.detach() * 2.0)
            s2, _ = pred(x)
            s2_ref, _ = pred(x, force_fp32=True)
        assert float((s2 - s2_ref).abs().max()) <= 2e-5 * max(1.0, float(s2_ref.abs().max()))
        if not norm:
            assert float((s2 - 2.0 * s_split).abs().max()) <= 1e-4 * max(1.0, float(s2.abs().max()))


def test_big_tile_segmean_matches_the_128x128_form(ops):
    """The mean-fused last convolution on the 256 x 256 tile (per-64-row-chunk column sums, fixed-order finish) against the
    128 x 128 form and against GEMM + mean in float64: ROI-major rows and residual (fp32 and split layout), ragged last tile,
    seg = 49 and 64.  The two kernels group a ROI's rows differently (64-row chunks vs 128-row tiles), so they agree to fp32
    summation noise, not bit for bit."""
    import os
    prev = os.environ.get("LOCOV_SPLIT_BIG")
    try:
        for R, seg, N, K, res_split in ((700, 49, 512, 128, False), (1111, 49, 256, 64, True), (300, 64, 264, 192, False), (53, 49, 2048, 512, True)):
            g = torch.Generator().manual_seed(R + N)
            M = R * seg
            x = torch.relu(torch.randn(M, K, generator=g)).cuda()
            xs = ops.split_pack(x, 16.0).data
            w = (torch.randn(N, K, generator=g) * 0.05).cuda()
            wp = ops.split_pack(w)
            res = torch.relu(torch.randn(M, N, generator=g)).cuda()
            if res_split:
                res_arg = ops.split_pack(res, 16.0).data
                res = ops.split_unpack(res_arg, 16.0)                  # what the kernels add: hi + lo
            else:
                res_arg = res
            sh = torch.randn(N, generator=g).cuda()
            outs = []
            for big in ("0", "1"):
                os.environ["LOCOV_SPLIT_BIG"] = big
                outs.append(ops.linear_split_segmean(xs, wp, sh, res_arg, seg, relu=True, residual_roi_major=True, x_is_split=True,
                                                     x_scale=16.0, residual_is_split=res_split))
            want = torch.relu(x.double() @ w.double().t() + sh.double() + res.double()).view(R, seg, N).mean(dim=1)
            scale = float(want.abs().max())
            for o in outs:
                assert tuple(o.shape) == (R, N)
                assert float((o.double() - want).abs().max()) <= 4e-6 * scale, (R, seg, N, K, res_split)
            assert float((outs[0] - outs[1]).abs().max()) <= 2e-6 * scale
            # deterministic
            assert torch.equal(outs[1], ops.linear_split_segmean(xs, wp, sh, res_arg, seg, relu=True, residual_roi_major=True, x_is_split=True,
                                                                 x_scale=16.0, residual_is_split=res_split))
    finally:
        if prev is None:
            os.environ.pop("LOCOV_SPLIT_BIG", None)
        else:
            os.environ["LOCOV_SPLIT_BIG"] = prev


def test_conv1_with_the_winograd_input_transform_in_its_epilogue_is_bit_identical(ops):
    """ops.conv1x1_winograd_conv3x3 (a bottleneck's conv1 + FrozenBN + ReLU and conv2 in one call): where the launch qualifies,
    conv1's epilogue on the 256 x 256 tile applies the Winograd input transform itself (gemm_split_big.hip MODE_WINO: tiles of 5
    whole ROIs, the finished half tile laid out in LDS, (ROI, fy) units shared by the 8 waves) and the pixel tensor between the
    two convolutions is never written.  Same BITS as the two separate calls (linear_split -> winograd_conv3x3), with ROI counts
    that leave a ragged last tile, fp32 and split-layout outputs, both output row orders; the range guard fires in both forms.
    LOCOV_WINO_FUSE=0 / LOCOV_SPLIT_BIG=1 (read per launch) select the form."""
    import os
    prev = {k: os.environ.get(k) for k in ("LOCOV_SPLIT_BIG", "LOCOV_WINO_FUSE")}
    try:
        for R, K, C, N, out_split, roi_major in ((1, 64, 256, 64, False, True), (5, 128, 256, 32, True, True), (23, 64, 512, 96, False, False),
                                                 (64, 192, 256, 64, True, True), (1307, 256, 256, 64, False, True)):
            g = torch.Generator().manual_seed(R + K)
            xs = ops.split_pack(torch.relu(torch.randn(49 * R, K, generator=g)).cuda(), 16.0).data
            w1 = ops.split_pack((torch.randn(C, K, generator=g) * 0.05).cuda())
            s1, b1 = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
            u = ops.split_pack(ops.winograd_pack_weight((torch.randn(N, C, 3, 3, generator=g) * 0.03).cuda()))
            s2, b2 = (torch.rand(N, generator=g) + 0.5).cuda(), (torch.randn(N, generator=g) * 0.1).cuda()
            oss = 16.0 if out_split else None
            os.environ["LOCOV_SPLIT_BIG"] = "0"
            y1 = ops.linear_split(xs, w1, b1, scale=s1, relu=True, x_is_split=True, x_scale=16.0)
            want = ops.winograd_conv3x3(y1, u, scale=s2, shift=b2, relu=True, roi_major=roi_major, in_roi_major=True, out_split_scale=oss)
            os.environ["LOCOV_SPLIT_BIG"] = "1"
            got = {}
            for fuse in ("0", "1"):
                os.environ["LOCOV_WINO_FUSE"] = fuse
                got[fuse] = ops.conv1x1_winograd_conv3x3(xs, w1, b1, u, scale1=s1, scale2=s2, shift2=b2, relu=True, x_scale=16.0,
                                                         roi_major=roi_major, out_split_scale=oss)
            assert torch.equal(got["0"], want), (R, K, C, N, "unfused form of the one-call entry point")
            assert torch.equal(got["1"], want), (R, K, C, N, "fused epilogue")
        # range guard: transform-domain values past fp16's range raise the word in the fused form too
        ops.split_overflow_reset(xs.device)
        big = ops.split_pack(torch.full((49 * 5, 64), 60.0).cuda(), 16.0).data
        wbig = ops.split_pack(torch.ones(256, 64).cuda())
        u1 = ops.split_pack(ops.winograd_pack_weight((torch.randn(32, 256, 3, 3, generator=torch.Generator().manual_seed(1)) * 0.03).cuda()))
        os.environ["LOCOV_WINO_FUSE"] = "1"
        ops.conv1x1_winograd_conv3x3(big, wbig, None, u1, x_scale=16.0, v_scale=16.0)            # pixels 3840, V up to 100x that, x 16
        assert ops.split_overflow_raised(xs.device)
        ops.split_overflow_reset(xs.device)
    finally:
        for k, v in prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_pooler_with_the_winograd_input_transform_is_bit_identical(ops):
    """ops.roi_align_winograd_conv3x3 (block 0: even-bin ROIAlign of the conv1-on-the-map channels + FrozenBN + ReLU, then conv2 in
    the Winograd domain): the ROIAlign workgroup (one ROI x 64 channels) keeps its 49 pooled rows in LDS and writes their input
    transform itself.  Same BITS as roi_align_nhwc -> winograd_conv3x3 and as the entry point's own two-launch form
    (LOCOV_WINO_FUSE=0), on a channel slice of a wider map, with boxes that leave the image, adaptive and fixed sampling grids."""
    import os
    prev = os.environ.get("LOCOV_WINO_FUSE")
    try:
        for Nimg, H, W, C, Cw, R, N2, sr, out_split in ((2, 25, 38, 64, 64, 37, 32, 0, False), (1, 50, 67, 128, 320, 101, 64, 2, True),
                                                       (3, 19, 23, 512, 2560, 64, 96, 0, False),
                                                       # more than 8 channel slices with a RAGGED last pass of the slice -> XCD
                                                       # mapping: 576 = 9 x 64, 1536 = 12 x 128
                                                       (2, 13, 17, 576, 576, 29, 32, 0, False), (1, 11, 15, 1536, 1600, 23, 32, 2, True)):
            g = torch.Generator().manual_seed(Nimg * 100 + C)
            fmap = torch.randn(Nimg, H, W, Cw, generator=g).cuda()
            feat = fmap[..., :C]
            wh = torch.rand(R, 2, generator=g) * torch.tensor([W * 16.0, H * 16.0]) * 0.9 + 4.0
            xy = torch.rand(R, 2, generator=g) * torch.tensor([W * 16.0, H * 16.0]) - 20.0
            rois = torch.cat([torch.randint(0, Nimg, (R, 1), generator=g).float(), xy, xy + wh], dim=1).cuda()
            s1, b1 = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
            u = ops.split_pack(ops.winograd_pack_weight((torch.randn(N2, C, 3, 3, generator=g) * 0.03).cuda()))
            s2, b2 = (torch.rand(N2, generator=g) + 0.5).cuda(), (torch.randn(N2, generator=g) * 0.1).cuda()
            oss = 16.0 if out_split else None
            y1 = ops.roi_align_nhwc(feat, rois, 14, 1.0 / 16, sr, True, bin_stride=2, ch_scale=s1, ch_shift=b1, relu=True).view(49 * R, C)
            want = ops.winograd_conv3x3(y1, u, scale=s2, shift=b2, relu=True, roi_major=True, in_roi_major=True, out_split_scale=oss)
            got = {}
            for fuse in ("0", "1"):
                os.environ["LOCOV_WINO_FUSE"] = fuse
                got[fuse] = ops.roi_align_winograd_conv3x3(feat, rois, 14, 1.0 / 16, sr, True, u, ch_scale=s1, ch_shift=b1, scale2=s2,
                                                           shift2=b2, relu=True, out_split_scale=oss)
            assert torch.equal(got["0"], want), (C, R, "two-launch form of the one-call entry point")
            assert torch.equal(got["1"], want), (C, R, "fused")
    finally:
        if prev is None:
            os.environ.pop("LOCOV_WINO_FUSE", None)
        else:
            os.environ["LOCOV_WINO_FUSE"] = prev
