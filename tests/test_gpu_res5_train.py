"""GPU parity of the Res5 stage under autograd (roi_emb_heads.py:323,343-347 with trainable Res5 convolutions,
configs/coco_lsm.yaml:8): the hand-written forward / data-gradient / weight-gradient kernels against a float64
torch-autograd evaluation of the oracle's bottleneck chain (oracle.bottleneck, oracle/lsm_oracle.py) on the same inputs.
Gate: every gradient within 1e-4 of the float64 one relative to that gradient's largest entry."""
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


def rel_err(got: torch.Tensor, want: torch.Tensor) -> float:
    want = want.double().cpu()
    return float((got.detach().double().cpu() - want).abs().max() / want.abs().max().clamp_min(1e-30))


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("M,N,K", [(1000, 128, 128), (777, 64, 96), (4097, 256, 36), (31, 4, 2048), (39200, 512, 256)])
def test_gemm_tn_vs_float64(pkg, M, N, K):
    ops = pkg.ops
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, N, generator=g).cuda()
    b = torch.randn(M, K, generator=g).cuda()
    s = (torch.rand(N, generator=g) + 0.5).cuda()
    want = (a.double().t() @ b.double()) * s.double()[:, None]
    got = ops.gemm_tn(a, b, s)
    assert rel_err(got, want) < 2e-6
    assert torch.equal(got, ops.gemm_tn(a, b, s)), "the chunk reduction must be deterministic"
    # strided operands (column blocks of wider matrices), no scale
    wide_a = torch.randn(M, N + 8, generator=g).cuda()
    wide_b = torch.randn(M, K + 12, generator=g).cuda()
    got = ops.gemm_tn(wide_a[:, 4:4 + N], wide_b[:, 8:8 + K])
    want = wide_a[:, 4:4 + N].double().t() @ wide_b[:, 8:8 + K].double()
    assert rel_err(got, want) < 2e-6


def test_gemm_tn_empty(pkg):
    ops = pkg.ops
    out = ops.gemm_tn(torch.zeros(0, 8, device="cuda"), torch.zeros(0, 12, device="cuda"))
    assert out.shape == (8, 12) and float(out.abs().max()) == 0.0


def test_linear_ex_mask_and_strided_weight(pkg):
    ops = pkg.ops
    g = torch.Generator().manual_seed(5)
    for (M, N, K) in ((300, 96, 64), (1000, 260, 128), (129, 36, 40)):
        x = torch.randn(M, K, generator=g).cuda()
        wide = torch.randn(N, K + 16, generator=g).cuda()
        w = wide[:, 8:8 + K]
        res = torch.randn(M, N, generator=g).cuda()
        act = torch.randn(M, N, generator=g).cuda()
        sc = (torch.rand(N, generator=g) + 0.5).cuda()
        want = (x.double() @ w.double().t()) * sc.double() + res.double()
        want = torch.where(act.double() > 0, want, torch.zeros_like(want))
        got = ops.linear_ex(x, w, scale=sc, residual=res, mask=act)
        assert rel_err(got, want) < 2e-6
        assert float(got[act <= 0].abs().max()) == 0.0
        # without a mask the extended entry equals the plain one bit for bit
        assert torch.equal(ops.linear_ex(x, w.contiguous(), scale=sc, residual=res), ops.linear(x, w.contiguous(), scale=sc, residual=res))


def test_small_backward_kernels(pkg):
    ops = pkg.ops
    g = torch.Generator().manual_seed(6)
    w = torch.randn(70, 52, generator=g).cuda()
    s = torch.randn(70, generator=g).cuda()
    assert torch.equal(ops.weight_transpose_scale(w, s), (w * s[:, None]).t().contiguous())
    assert torch.equal(ops.weight_transpose_scale(w), w.t().contiguous())
    w3 = torch.randn(24, 40, 3, 3, generator=g).cuda()
    s3 = torch.randn(24, generator=g).cuda()
    want = (w3 * s3[:, None, None, None]).flip(2, 3).permute(1, 0, 2, 3).contiguous()
    assert torch.equal(ops.conv3x3_weight_flip(w3, s3), want)
    # im2col == unfold (tap-major columns)
    x = torch.randn(3, 8, 5, 6, generator=g).cuda()
    rows = x.permute(0, 2, 3, 1).reshape(-1, 8).contiguous()
    col = ops.im2col3x3(rows, 5, 6)
    want = F.unfold(x, 3, padding=1).view(3, 8, 9, 30).permute(0, 3, 2, 1).reshape(90, 72)
    assert torch.equal(col, want.contiguous())
    p = torch.randn(24, 9 * 40, generator=g).cuda()
    want = (p.view(24, 9, 40).permute(0, 2, 1).reshape(24, 40, 3, 3) * s3[:, None, None, None]).contiguous()
    assert torch.equal(ops.conv3x3_wgrad_unpack(p, s3), want)
    a, gr = torch.randn(50, 12, generator=g).cuda(), torch.randn(50, 12, generator=g).cuda()
    assert torch.equal(ops.relu_mask(gr, a), torch.where(a > 0, gr, torch.zeros_like(gr)))
    gp = torch.randn(7, 12, generator=g).cuda()
    act = torch.randn(7 * 9, 12, generator=g).cuda()
    want = torch.where(act > 0, (gp * (1.0 / 9.0)).repeat_interleave(9, 0), torch.zeros_like(act))
    torch.testing.assert_close(ops.spatial_mean_bwd(gp, act, 9), want, rtol=1e-6, atol=0)
    m = torch.randn(2, 5, 7, 8, generator=g).cuda()          # [N,H,W,C], odd sizes
    assert torch.equal(ops.rows_stride2(m, 2, 5, 7, True).view(2, 3, 4, 8), m[:, ::2, ::2])
    r = torch.randn(2 * 3 * 4, 8, generator=g).cuda()
    back = ops.rows_stride2(r, 2, 5, 7, False)
    want = torch.zeros_like(m)
    want[:, ::2, ::2] = r.view(2, 3, 4, 8)
    assert torch.equal(back, want)
    assert torch.equal(ops.nhwc_to_nchw(m), m.permute(0, 3, 1, 2).contiguous())


def test_roi_align_even_backward_is_the_adjoint(pkg, oracle):
    """<ROIAlign_even(F), G> == <F, ROIAlign_even^T(G)>, and the backward equals the oracle's torchvision-style
    roi_align_backward fed the gradient at the even bins only."""
    ops = pkg.ops
    rng = np.random.default_rng(8)
    N, C, H, W, R = 2, 32, 50, 84, 60
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    boxes = [oracle.synth_boxes(rng, R // 2), oracle.synth_boxes(rng, R // 2)]
    boxes[0][0] = [-40.0, -30.0, 20.0, 25.0]           # partly outside the map
    boxes[1][1] = [5.0, 5.0, 5.0, 5.0]                 # empty box
    rois = oracle.boxes_to_pooler_format(boxes)
    nhwc = ops.nchw_to_nhwc(dev(feat))
    for pos_major in (False, True):
        y = ops.roi_align_nhwc(nhwc, dev(rois), 14, 1 / 16, 0, True, bin_stride=2, pos_major=pos_major)
        G = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).cuda()
        gf = ops.roi_align_nhwc_bwd(G.view(-1, C), (N, H, W, C), dev(rois), 14, 1 / 16, 0, True, bin_stride=2, pos_major=pos_major)
        lhs = float((y.double() * G.double()).sum())
        rhs = float((nhwc.double() * gf.double()).sum())
        assert abs(lhs - rhs) <= 1e-5 * float(y.double().norm() * G.double().norm())
        if not pos_major:
            g14 = np.zeros((R, C, 14, 14), np.float32)
            g14[:, :, ::2, ::2] = G.view(R, 7, 7, C).permute(0, 3, 1, 2).cpu().numpy()
            want = oracle.roi_align_backward(g14, (N, C, H, W), rois, 1 / 16, 0, True)
            got = gf.permute(0, 3, 1, 2).cpu().numpy()
            assert np.abs(got - want).max() <= 1e-5 * max(np.abs(want).max(), 1.0)


def test_roi_align_even_backward_window_equals_the_direct_scatter(pkg, oracle, monkeypatch):
    """The backward of the even-grid ROIAlign adds a SMALL proposal's bins up per pixel in an LDS window before it touches memory
    (one atomic per touched pixel and channel instead of one per bin and pixel); larger proposals scatter directly.  At the LSM
    step's shape (800 proposals over 4 images, 1024 channels, the bench's size mix incl. boxes that leave the image and empty ones)
    both forms give the same map gradient up to the order of the fp32 additions, and a float64 evaluation of the adjoint agrees."""
    ops = pkg.ops
    rng = np.random.default_rng(18)
    N, C, H, W, per = 4, 1024, 50, 84, 200
    boxes = [oracle.synth_boxes(rng, per) for _ in range(N)]
    boxes[0][0] = [-40.0, -30.0, 20.0, 25.0]
    boxes[1][1] = [5.0, 5.0, 5.0, 5.0]
    boxes[2][:40] = np.concatenate([rng.uniform(0, 1200, (40, 2)), rng.uniform(0, 1200, (40, 2))], 1).astype(np.float32)
    boxes[2][:40, 2:] = boxes[2][:40, :2] + rng.uniform(8, 60, (40, 2)).astype(np.float32)      # many tiny proposals
    rois = dev(oracle.boxes_to_pooler_format(boxes))
    G = torch.randn(49 * N * per, C, generator=torch.Generator().manual_seed(2)).cuda()
    out = {}
    monkeypatch.setenv("LOCOV_POOL_BWD_TILES", "0")          # (the scatter forms; the ownership form has its own test below)
    for win in ("0", "20480", "65536"):
        monkeypatch.setenv("LOCOV_POOL_BWD_WINDOW", win)
        out[win] = ops.roi_align_nhwc_bwd(G, (N, H, W, C), rois, 14, 1 / 16, 0, True, bin_stride=2)
    monkeypatch.delenv("LOCOV_POOL_BWD_WINDOW")
    monkeypatch.delenv("LOCOV_POOL_BWD_TILES")
    scale = float(out["0"].abs().max())
    assert scale > 0
    for win in ("20480", "65536"):
        assert float((out[win] - out["0"]).abs().max()) <= 2e-6 * scale, win
    # the adjoint identity in float64 on a channel slice: <ROIAlign_even(F), G> == <F, ROIAlign_even^T(G)>
    F = torch.randn(N, H, W, C, generator=torch.Generator().manual_seed(3)).cuda()
    y = ops.roi_align_nhwc(F, rois, 14, 1 / 16, 0, True, bin_stride=2).view(-1, C)
    lhs, rhs = float((y.double() * G.double()).sum()), float((F.double() * out["20480"].double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * float(y.double().norm() * G.double().norm())


@pytest.mark.parametrize("N,C,H,W,per", [(4, 1024, 50, 84, 200), (3, 1024, 50, 84, 512), (2, 128, 13, 9, 40), (1, 256, 64, 64, 2500),
                                         (2, 128, 50, 84, 6000)])        # (12 000 proposals: six list passes per tile -- ADVICE r5)
def test_roi_align_even_backward_by_tile_ownership(pkg, oracle, N, C, H, W, per, monkeypatch):
    """The ownership form of the even-grid ROIAlign backward (a workgroup per 8 x 8 map tile and 128-channel slice collects the
    proposals that reach its tile; no atomics) against the scatter form at the LSM / STT steps' shapes, a map smaller than two tiles
    and more proposals than one list pass holds: the same gradient up to the order of the fp32 additions, the oracle's
    roi_align_backward on a channel slice, accumulation into an existing gradient -- and the same bits on every run."""
    ops = pkg.ops
    rng = np.random.default_rng(18 + per)
    boxes = [oracle.synth_boxes(rng, per) for _ in range(N)]
    for b in boxes:
        b[:, [0, 2]] *= (W * 16) / 1333.0
        b[:, [1, 3]] *= (H * 16) / 800.0
    boxes[0][0] = [-40.0, -30.0, 20.0, 25.0]                                     # partly outside the map
    boxes[-1][1] = [5.0, 5.0, 5.0, 5.0]                                          # empty box
    boxes[0][2] = [0.0, 0.0, W * 16.0 + 50.0, H * 16.0 + 50.0]                   # larger than the map
    boxes[-1][3:23] = np.concatenate([rng.uniform(0, W * 12, (20, 2)), np.zeros((20, 2))], 1).astype(np.float32)
    boxes[-1][3:23, 2:] = boxes[-1][3:23, :2] + rng.uniform(8, 60, (20, 2)).astype(np.float32)     # tiny proposals
    rois_np = oracle.boxes_to_pooler_format(boxes)
    rois = dev(rois_np)
    G = torch.randn(49 * N * per, C, generator=torch.Generator().manual_seed(2)).cuda()
    monkeypatch.setenv("LOCOV_POOL_BWD_TILES", "0")
    want = ops.roi_align_nhwc_bwd(G, (N, H, W, C), rois, 14, 1 / 16, 0, True, bin_stride=2)
    monkeypatch.delenv("LOCOV_POOL_BWD_TILES")
    got = ops.roi_align_nhwc_bwd(G, (N, H, W, C), rois, 14, 1 / 16, 0, True, bin_stride=2)
    scale = float(want.abs().max())
    assert scale > 0 and float((got - want).abs().max()) <= 3e-6 * scale
    assert torch.equal(got, ops.roi_align_nhwc_bwd(G, (N, H, W, C), rois, 14, 1 / 16, 0, True, bin_stride=2))      # reproducible
    # the oracle's torchvision-style backward fed the gradient at the even bins, on the first 8 channels
    R = N * per
    if R <= 1600:
        g14 = np.zeros((R, 8, 14, 14), np.float32)
        g14[:, :, ::2, ::2] = G[:, :8].view(R, 7, 7, 8).permute(0, 3, 1, 2).cpu().numpy()
        ref = oracle.roi_align_backward(g14, (N, 8, H, W), rois_np, 1 / 16, 0, True)
        assert np.abs(got[..., :8].permute(0, 3, 1, 2).cpu().numpy() - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1.0)
    # accumulation into an existing gradient, a proposal whose image index is out of range (ignored), no proposals at all
    seed = torch.randn(N, H, W, C, generator=torch.Generator().manual_seed(4)).cuda()
    acc = ops.roi_align_nhwc_bwd(G, (N, H, W, C), rois, 14, 1 / 16, 0, True, bin_stride=2, accumulate_into=seed.clone())
    assert float((acc - (seed + got)).abs().max()) <= 1e-6 * max(scale, 1.0)
    bad = rois.clone()
    bad[:5, 0] = N + 3
    g_bad = ops.roi_align_nhwc_bwd(G, (N, H, W, C), bad, 14, 1 / 16, 0, True, bin_stride=2)
    G0 = G.clone()
    G0.view(R, 49, C)[:5] = 0
    assert float((g_bad - ops.roi_align_nhwc_bwd(G0, (N, H, W, C), rois, 14, 1 / 16, 0, True, bin_stride=2)).abs().max()) <= 3e-6 * scale
    assert not bool(ops.roi_align_nhwc_bwd(G[:0], (N, H, W, C), rois[:0], 14, 1 / 16, 0, True, bin_stride=2).any())


@pytest.mark.parametrize("R,Cin,N", [(5, 32, 48), (70, 64, 64), (300, 128, 96)])
def test_winograd_gradients_vs_float64(pkg, R, Cin, N):
    """3x3 convolution of 7x7 tiles: weight gradient (Winograd-domain TN GEMMs) and masked data gradient (Winograd
    convolution with the flipped filter) against float64 autograd of F.conv2d."""
    ops = pkg.ops
    gen = torch.Generator().manual_seed(R)
    x = torch.randn(R, Cin, 7, 7, generator=gen)
    w = torch.randn(N, Cin, 3, 3, generator=gen) * 0.1
    s = torch.rand(N, generator=gen) + 0.5
    gy = torch.randn(R, N, 7, 7, generator=gen)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    (F.conv2d(xd, wd, padding=1) * s.double().view(1, -1, 1, 1) * gy.double()).sum().backward()
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(R * 49, -1).contiguous().cuda()
    dw = ops.winograd_wgrad(rows(x), rows(gy), s.cuda(), roi_major=True)
    assert rel_err(dw, wd.grad) < 2e-5
    # position-major rows give the same gradient
    prow = lambda t: t.permute(2, 3, 0, 1).reshape(49 * R, -1).contiguous().cuda()
    dwp = ops.winograd_wgrad(prow(x), prow(gy), s.cuda(), roi_major=False)
    assert rel_err(dwp, wd.grad) < 2e-5
    if N % 32 == 0:           # the data gradient is a convolution FROM the N output channels
        act = torch.randn(R, Cin, 7, 7, generator=gen)
        U = ops.winograd_pack_weight(ops.conv3x3_weight_flip(w.cuda(), s.cuda()))
        gx = ops.winograd_conv3x3_ex(rows(gy), U, mask=rows(act), roi_major=True)
        want = torch.where(act.double() > 0, xd.grad, torch.zeros_like(xd.grad))
        assert rel_err(gx.view(R, 7, 7, Cin).permute(0, 3, 1, 2), want) < 2e-5


@pytest.mark.parametrize("R,H,W,Cin,N", [(3, 13, 21, 32, 40), (2, 25, 42, 64, 64)])
def test_grid_conv3x3_gradients_vs_float64(pkg, R, H, W, Cin, N):
    """The general-grid form: weight gradient = TN GEMM against im2col patches, data gradient = implicit-GEMM convolution
    with the flipped filter and the mask epilogue."""
    ops = pkg.ops
    gen = torch.Generator().manual_seed(H)
    x = torch.randn(R, Cin, H, W, generator=gen)
    w = torch.randn(N, Cin, 3, 3, generator=gen) * 0.1
    s = torch.rand(N, generator=gen) + 0.5
    gy = torch.randn(R, N, H, W, generator=gen)
    act = torch.randn(R, Cin, H, W, generator=gen)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    (F.conv2d(xd, wd, padding=1) * s.double().view(1, -1, 1, 1) * gy.double()).sum().backward()
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(R * H * W, -1).contiguous().cuda()
    dw = ops.conv3x3_wgrad_unpack(ops.gemm_tn(rows(gy), ops.im2col3x3(rows(x), H, W)), s.cuda())
    assert rel_err(dw, wd.grad) < 2e-6
    if N % 32 == 0:
        wfp = ops.pack_conv3x3_weight(ops.conv3x3_weight_flip(w.cuda(), s.cuda()))
        gx = ops.conv3x3_nhwc_ex(rows(gy), wfp, H, W, mask=rows(act))
        want = torch.where(act.double() > 0, xd.grad, torch.zeros_like(xd.grad))
        assert rel_err(gx.view(R, H, W, Cin).permute(0, 3, 1, 2), want) < 2e-6


# ------------------------------------------------------------------------------------------------ the stage
def _stage(pkg, oracle, in_ch, mid, out_ch, seed):
    from locov_amd.config import get_cfg
    from locov_amd.res5 import build_res5_block
    cfg = get_cfg()
    cfg.MODEL.RESNETS.RES2_OUT_CHANNELS = out_ch // 8
    cfg.MODEL.RESNETS.WIDTH_PER_GROUP = mid // 8
    res5, oc = build_res5_block(cfg)
    assert oc == out_ch
    params = oracle.make_res5_params(seed, in_ch=in_ch, mid=mid, out_ch=out_ch)
    res5.load_state_dict(params)
    return res5.cuda().train(), params


def _float64_stage(oracle, params, x, masks=None, p=None):
    """The oracle's bottleneck chain (oracle.bottleneck = [D2-upstream] BottleneckBlock.forward) with autograd on, in
    float64.  masks = None: oracle.bottleneck itself.  masks = [(y1 > 0, y2 > 0, out > 0)] per block, NCHW bool: the same
    chain with every ReLU replaced by the given 0/1 pattern.

    Why: a ReLU's derivative is a step function.  A pre-activation that is zero to within fp32 rounding can land on either
    side in an fp32 forward and in the float64 one; such a unit passes its whole gradient in one evaluation and none in the
    other, so ONE flipped unit moves entries of a weight gradient by ~1e-3 of the largest entry (measured: 1 of 263 000 units
    of the stage output flips at R = 21, and d/dW of the last convolution then differs by 2e-3) -- any two correct fp32
    implementations differ from float64, and from each other, by that much.  The gradient gate is therefore evaluated on the
    function the device actually computed: float64 arithmetic, the device's active set.  The forward (and hence the active
    set, up to such ties) is gated separately against the plain oracle."""
    if p is None:                                         # (p: the float64 parameters of an earlier call -- gradients then accumulate)
        p = {k: v.double().requires_grad_(".norm." not in k) for k, v in params.items()}
    y = x
    for i, stride in enumerate((2, 1, 1)):
        pre = f"{i}."
        if masks is None:
            y = oracle.bottleneck(y, p, pre, stride, True)
            continue
        m1, m2, m3 = (m.double() for m in masks[i])
        o = oracle.frozen_bn(F.conv2d(y, p[pre + "conv1.weight"], stride=stride), p, pre + "conv1.norm.") * m1
        o = oracle.frozen_bn(F.conv2d(o, p[pre + "conv2.weight"], padding=1), p, pre + "conv2.norm.") * m2
        o = oracle.frozen_bn(F.conv2d(o, p[pre + "conv3.weight"]), p, pre + "conv3.norm.")
        sc = y
        if (pre + "shortcut.weight") in p:
            sc = oracle.frozen_bn(F.conv2d(y, p[pre + "shortcut.weight"], stride=stride), p, pre + "shortcut.norm.")
        y = (o + sc) * m3
    return y, p


def _device_masks(fn_out, R, H, W):
    """The active sets of the device forward: (y1 > 0, y2 > 0, out > 0) per block as NCHW bool tensors on the CPU, read from
    the activations the per-block Res5 nodes saved for their backward (res5_train.saved_activations)."""
    from locov_amd import res5_train
    saved = res5_train.saved_activations(fn_out)
    assert saved and len(saved) % 4 == 0
    nchw = lambda t: (t.detach().view(R, H, W, -1).permute(0, 3, 1, 2) > 0).cpu()
    return [tuple(nchw(saved[4 * b + j]) for j in (1, 2, 3)) for b in range(len(saved) // 4)]


def test_float64_reference_with_its_own_masks_is_the_oracle(oracle):
    """The masked restatement used by the gradient gates equals oracle.bottleneck when given the oracle's own active sets."""
    params = oracle.make_res5_params(3, in_ch=32, mid=16, out_ch=64)
    x = torch.randn(3, 32, 14, 14, generator=torch.Generator().manual_seed(2)).double()
    y, _ = _float64_stage(oracle, params, x)
    masks, t = [], x
    p = {k: v.double() for k, v in params.items()}
    for i, stride in enumerate((2, 1, 1)):
        pre = f"{i}."
        o1 = F.relu(oracle.frozen_bn(F.conv2d(t, p[pre + "conv1.weight"], stride=stride), p, pre + "conv1.norm."))
        o2 = F.relu(oracle.frozen_bn(F.conv2d(o1, p[pre + "conv2.weight"], padding=1), p, pre + "conv2.norm."))
        t = oracle.bottleneck(t, p, pre, stride, True)
        masks.append((o1 > 0, o2 > 0, t > 0))
    ym, _ = _float64_stage(oracle, params, x, masks)
    assert torch.equal(y.detach(), ym.detach())


def _weight_keys(params):
    return [k for k in params if k.endswith(".weight") and ".norm." not in k]


@pytest.mark.parametrize("dims,R,split", [((128, 64, 256), 21, False), ((128, 64, 256), 21, True),
                                          ((1024, 512, 2048), 12, False), ((1024, 512, 2048), 12, True)])
def test_res5_rows_gradients_vs_float64(pkg, oracle, dims, R, split):
    """ROI tiles (7x7 after block 0's stride): d loss / d {stage input, every convolution weight} with the spatial mean
    behind the stage (roi_emb_heads.py:343-344), small and configs/coco_lsm.yaml channel sizes, both forward arithmetics."""
    from locov_amd import res5_train
    in_ch, mid, out_ch = dims
    res5, params = _stage(pkg, oracle, in_ch, mid, out_ch, seed=R)
    gen = torch.Generator().manual_seed(17)
    x14 = torch.randn(R, in_ch, 14, 14, generator=gen)
    gy = torch.randn(R, out_ch, generator=gen)
    x0 = x14[:, :, ::2, ::2].permute(0, 2, 3, 1).reshape(R * 49, in_ch).contiguous().cuda().requires_grad_(True)
    out = res5_train.res5_rows(res5, x0, R, 7, 7, pooled=True, split=split)
    with torch.no_grad():
        y_oracle, _ = _float64_stage(oracle, params, x14.double())
    assert rel_err(out, y_oracle.mean(dim=[2, 3])) < (2e-5 if split else 1e-5)        # forward: the plain oracle
    masks = _device_masks(out, R, 7, 7)
    flips = int((masks[-1][2] != (y_oracle > 0)).sum())
    assert flips <= 1e-4 * y_oracle.numel(), flips                                       # active sets agree up to fp32 ties
    xd = x14.double().requires_grad_(True)
    yd, pd = _float64_stage(oracle, params, xd, masks)
    (yd.mean(dim=[2, 3]) * gy.double()).sum().backward()
    (out * gy.cuda()).sum().backward()
    want_x = xd.grad[:, :, ::2, ::2].permute(0, 2, 3, 1).reshape(R * 49, in_ch)
    assert float(xd.grad[:, :, 1::2, :].abs().max()) == 0.0       # stride 2: the odd positions carry no gradient
    errs = {"x0": rel_err(x0.grad, want_x)}
    sd = dict(res5.named_parameters())
    for k in _weight_keys(params):
        errs[k] = rel_err(sd[k].grad, pd[k].grad)
    assert max(errs.values()) < 1e-4, errs
    # un-pooled output + plain (non-mean) gradient
    x0b = x0.detach().clone().requires_grad_(True)
    res5.zero_grad()
    rows = res5_train.res5_rows(res5, x0b, R, 7, 7, pooled=False, split=split)
    G = torch.randn(rows.shape, generator=gen).cuda()
    masks = _device_masks(rows, R, 7, 7)
    (rows * G).sum().backward()
    xd2 = x14.double().requires_grad_(True)
    yd2, pd2 = _float64_stage(oracle, params, xd2, masks)
    (yd2 * G.view(R, 7, 7, out_ch).permute(0, 3, 1, 2).double().cpu()).sum().backward()
    errs = {"x0": rel_err(x0b.grad, xd2.grad[:, :, ::2, ::2].permute(0, 2, 3, 1).reshape(R * 49, in_ch))}
    for k in _weight_keys(params):
        errs[k] = rel_err(sd[k].grad, pd2[k].grad)
    assert max(errs.values()) < 1e-4, errs


@pytest.mark.parametrize("one_launch", [False, True], ids=["packed_in_the_backward", "one_launch_prep"])
def test_a_stale_backward_weight_scale_never_reaches_the_parameters(pkg, oracle, one_launch, monkeypatch):
    """the split-arithmetic backward re-uses remembered power-of-two weight scales; when one stops covering its
    (grown) weight the pack kernel raises a range-guard word and the GEMMs behind it would produce inf / NaN.
    Operands packed inside the backward (the steps that choose scales; LOCOV_RES5_PREP=0): the stage's "bwd" word -- nothing
    reads it inside autograd, so the pass itself keeps those values from an optimizer: every gradient it produced is zero-filled
    on the device (locov_zero_if_raised), the word stays set for the next host read, and the step after that (scales chosen
    afresh) is correct again.  Operands from the step's ONE preparation launch (every other step): that launch runs under the
    FORWARD's guard, whose reader repeats the step on the f32 MFMA -- correct gradients at once -- and drops the scales."""
    from locov_amd import res5 as res5_mod, res5_train
    monkeypatch.setattr(res5_mod, "_ONE_LAUNCH_PREP", one_launch)
    R, (in_ch, mid, out_ch) = 21, (128, 64, 256)
    res5, params = _stage(pkg, oracle, in_ch, mid, out_ch, seed=5)
    gen = torch.Generator().manual_seed(23)
    x0 = torch.randn(R * 49, in_ch, generator=gen).cuda()
    gy = torch.randn(R, out_ch, generator=gen).cuda()
    named = dict(res5.named_parameters())
    keys = _weight_keys(params)

    def step():
        res5.zero_grad()
        x = x0.clone().requires_grad_(True)
        (res5_train.res5_rows(res5, x, R, 7, 7, pooled=True, split=True) * gy).sum().backward()
        torch.cuda.synchronize()
        return x.grad.clone(), {k: named[k].grad.clone() for k in keys}

    gx_ref, gw_ref = step()
    assert not res5.backward_guard_raised(x0.device) and float(gx_ref.abs().max()) > 0 and all(float(g.abs().max()) > 0 for g in gw_ref.values())
    # an "optimizer step" (every weight's version moves: all packings are redone) and remembered BACKWARD scales 2^14 too large
    with torch.no_grad():
        for k in keys:
            named[k].add_(0.0)
    poisoned = 0
    for key, (scale, _) in list(res5._scales.items()):
        if isinstance(key[1], str):                              # (id(conv), "t" / "uflip" / "flip9"): the backward's operands
            res5._scales[key] = (scale * 2.0 ** 14, 0)
            poisoned += 1
    assert poisoned >= 9
    gx, gw = step()
    if one_launch:
        # the f32 MFMA's results: two fp32 evaluations of one step (a ReLU tie may flip between them, see _float64_stage)
        rel_l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
        assert rel_l2(gx, gx_ref) < 5e-3 and all(rel_l2(gw[k], gw_ref[k]) < 5e-3 for k in keys)
        assert not res5.backward_guard_raised(x0.device) and not res5._scales
        gx2, gw2 = step()
        assert torch.equal(gx2, gx_ref) and all(torch.equal(gw2[k], gw_ref[k]) for k in keys)
        return
    for name, g in [("x0", gx)] + list(gw.items()):
        assert bool(torch.isfinite(g).all()), name
        assert float(g.abs().max()) == 0.0, f"{name}: the skipped pass must leave zeros, not partial gradients"
    assert res5.backward_guard_raised(x0.device)                    # ... and says so at the next host read (which drops the scales)
    assert not res5.backward_guard_raised(x0.device)
    gx2, gw2 = step()
    assert torch.equal(gx2, gx_ref) and all(torch.equal(gw2[k], gw_ref[k]) for k in keys)
    assert not res5.backward_guard_raised(x0.device)


def test_zero_if_raised(pkg):
    ops = pkg.ops
    g = torch.Generator().manual_seed(3)
    ts = [torch.randn(n, generator=g).cuda() for n in (1, 3, 4, 1000, 70001)] + [None, torch.zeros(0, device="cuda")]
    ts.append(torch.randn(16, 37, generator=g).cuda()[1:])       # not 16-byte aligned
    keep = [t.clone() if t is not None else None for t in ts]
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.zero_if_raised(ts, word)
    assert all(t is None or torch.equal(t, k) for t, k in zip(ts, keep))
    word.fill_(1)
    ops.zero_if_raised(ts + [torch.randn(5, generator=g).cuda() for _ in range(30)], word)       # more than one launch's list
    assert all(t is None or float(t.abs().sum()) == 0.0 for t in ts)
    with pytest.raises(Exception):
        ops.zero_if_raised([torch.zeros(4, 4, device="cuda").t()], word)


@pytest.mark.parametrize("split", [False, True], ids=["f32mfma", "f16x2"])
@pytest.mark.parametrize("dims,N,H,W", [((128, 64, 256), 2, 26, 43), ((1024, 512, 2048), 2, 50, 84)])
def test_res5_grid_gradients_vs_float64(pkg, oracle, dims, N, H, W, split):
    """roi_emb_heads.py:323: the stage on the whole res4 grid (NCHW in, NCHW out), output and gradients w.r.t. the map
    and every convolution weight -- on the f32 MFMA and in the DEFAULT split-operand arithmetic (forward, im2col 3x3, TN
    weight gradients and masked data gradients all on (hi, lo) f16 pairs), same gates."""
    from locov_amd import res5_train
    in_ch, mid, out_ch = dims
    res5, params = _stage(pkg, oracle, in_ch, mid, out_ch, seed=N + H)
    gen = torch.Generator().manual_seed(23)
    feat = torch.randn(N, in_ch, H, W, generator=gen)
    f = feat.cuda().requires_grad_(True)
    y = res5_train.res5_grid(res5, res5_train.to_nhwc(f), split=split)
    with torch.no_grad():
        y_oracle, _ = _float64_stage(oracle, params, feat.double())
    assert tuple(y.shape) == tuple(y_oracle.shape)
    assert rel_err(y, y_oracle) < 1e-5                                                   # forward: the plain oracle
    masks = _device_masks(y, N, y.shape[2], y.shape[3])
    assert int((masks[-1][2] != (y_oracle > 0)).sum()) <= 1e-4 * y_oracle.numel()
    fd = feat.double().requires_grad_(True)
    yd, pd = _float64_stage(oracle, params, fd, masks)
    G = torch.randn(yd.shape, generator=gen)
    (yd * G.double()).sum().backward()
    (y * G.cuda()).sum().backward()
    errs = {"feat": rel_err(f.grad, fd.grad)}
    sd = dict(res5.named_parameters())
    for k in _weight_keys(params):
        errs[k] = rel_err(sd[k].grad, pd[k].grad)
    assert max(errs.values()) < 1e-4, errs


@pytest.mark.parametrize("together", [False, True], ids=["forwards_per_segment", "forwards_together"])
@pytest.mark.parametrize("split", [False, True], ids=["f32mfma", "f16x2"])
def test_joint_step_equals_the_two_calls_and_float64(pkg, oracle, split, together):
    """res5_train.Res5Step: the whole-grid call (roi_emb_heads.py:323) and the proposals' call (:343-344) of one LSM step as the
    two segments of ONE autograd node -- every 1x1 data / weight gradient is one launch over the joint rows.  Outputs are
    bit-identical to the two separate calls (the forward IS per segment); every gradient (the map through both paths, every
    convolution weight as the sum of both calls' contributions) agrees with the two-call form and with a float64 evaluation on
    the device's own active sets, with very different gradient magnitudes on the two segments (the joint pass shares one
    operand scale per gradient matrix)."""
    from locov_amd import res5_train, ops
    in_ch, mid, out_ch = 128, 64, 256
    res5, params = _stage(pkg, oracle, in_ch, mid, out_ch, seed=77)
    gen = torch.Generator().manual_seed(41)
    N, H, W, R = 2, 26, 43, 19
    feat = torch.randn(N, in_ch, H, W, generator=gen)
    OH, OW = (H + 1) // 2, (W + 1) // 2
    wh = torch.rand(R, 2, generator=gen) * torch.tensor([W * 16.0, H * 16.0]) * 0.5 + 24.0
    xy = torch.rand(R, 2, generator=gen) * torch.tensor([W * 8.0, H * 8.0])
    rois = torch.cat([torch.randint(0, N, (R, 1), generator=gen).float(), xy, xy + wh], dim=1).cuda()
    Gg = torch.randn(N, out_ch, OH, OW, generator=gen).cuda() * 1e-3            # a grounding-loss-sized gradient on the grid ...
    Gr = torch.randn(R, out_ch, generator=gen).cuda() * 3.0                      # ... against a 3 000x larger one on the proposals
    sd = dict(res5.named_parameters())
    keys = _weight_keys(params)

    def two_calls():
        res5.zero_grad()
        f = feat.cuda().requires_grad_(True)
        nhwc = res5_train.to_nhwc(f)
        grid = res5_train.res5_grid(res5, nhwc, split=split)
        box = res5_train.res5_rois(res5, nhwc, rois, 14, 1.0 / 16, 0, True, pooled=True, split=split)
        ((grid * Gg).sum() + (box * Gr).sum()).backward()
        return grid.detach(), box.detach(), f.grad.clone(), {k: sd[k].grad.clone() for k in keys}

    def joint():
        res5.zero_grad()
        f = feat.cuda().requires_grad_(True)
        nhwc = res5_train.to_nhwc(f)
        step = res5_train.Res5Step(res5, split, f.device, N * OH * OW + 49 * R + 100)      # (spare capacity: fewer proposals than planned)
        if together:          # (the sample is known without a host wait: the 1x1 convolutions of both calls share launches)
            rows, x0 = res5_train.grid_and_roi_segments(step, nhwc, rois, 14, 1.0 / 16, 0, True)
        else:
            rows = res5_train.grid_segment(step, nhwc)
            x0 = res5_train.roi_segment(step, nhwc, rois, 14, 1.0 / 16, 0, True)
        grid_rows, box = step.outputs([rows, x0], [False, True])
        grid = res5_train.to_nchw(grid_rows, N, OH, OW)
        # the block nodes' own active sets (grid rows first, then the proposals' 7x7 tiles), read before the backward frees them
        saved = res5_train.saved_activations(grid)
        assert len(saved) == 12
        ng = N * OH * OW
        assert saved[0].shape[0] == ng + 49 * R                                  # the rows in use, not the capacity
        gm = [tuple((saved[4 * b + j][:ng].view(N, OH, OW, -1).permute(0, 3, 1, 2) > 0).cpu() for j in (1, 2, 3)) for b in range(3)]
        rm = [tuple((saved[4 * b + j][ng:].view(R, 7, 7, -1).permute(0, 3, 1, 2) > 0).cpu() for j in (1, 2, 3)) for b in range(3)]
        ((grid * Gg).sum() + (box * Gr).sum()).backward()
        return grid.detach(), box.detach(), f.grad.clone(), {k: sd[k].grad.clone() for k in keys}, gm, rm

    g2, b2, fx2, w2 = two_calls()
    gj, bj, fxj, wj, gmasks, rmasks = joint()
    assert torch.equal(gj, g2) and torch.equal(bj, b2)
    assert rel_err(fxj, fx2) < 1e-5
    for k in keys:
        assert rel_err(wj[k], w2[k]) < 1e-5, k
    # float64 on those active sets
    fd = feat.double().requires_grad_(True)
    yg, pd = _float64_stage(oracle, params, fd, gmasks)
    # the proposals' stage input in float64: the device's even-grid ROIAlign is linear in the map -- differentiate it through
    # the fp32 op's own adjoint by treating it as a fixed linear map: x14 = A f  (A applied by the device op on a float64-valued copy)
    x0d = ops.roi_align_nhwc(ops.nchw_to_nhwc(feat.cuda()), rois, 14, 1.0 / 16, 0, True, bin_stride=2).view(R, 7, 7, in_ch)
    x14 = torch.zeros(R, in_ch, 14, 14, dtype=torch.float64)
    x14[:, :, ::2, ::2] = x0d.permute(0, 3, 1, 2).double().cpu()
    x14.requires_grad_(True)
    yr, _ = _float64_stage(oracle, params, x14, rmasks, p=pd)
    ((yg * Gg.double().cpu()).sum() + (yr.mean(dim=[2, 3]) * Gr.double().cpu()).sum()).backward()
    for k in keys:
        assert rel_err(wj[k], pd[k].grad) < 1e-4, k


def test_one_launch_weight_prep_is_bit_identical(pkg):
    """locov_res5_weight_prep: every split-layout operand of a training step from ONE launch, against the multi-launch chain it
    replaces (transpose / flip / Winograd filter transform / im2col packing, then split_pack) -- the same bits, for Res5's shapes
    and a small one, with and without the FrozenBN row scale."""
    ops = pkg.ops
    g = torch.Generator().manual_seed(12)
    for (N, C) in ((64, 128), (512, 2048), (2048, 512), (96, 32)):
        w = (torch.randn(N, C, generator=g) * 0.05).cuda()
        s = (torch.rand(N, generator=g) + 0.5).cuda()
        plain, t = torch.empty(N, C, device="cuda"), torch.empty(C, N, device="cuda")
        ops.res5_weight_prep([("plain", w, None, plain, 4096.0), ("t", w, s, t, 2048.0)])
        assert torch.equal(plain.view(torch.int32), ops.split_pack(w, 4096.0).data.view(torch.int32))
        assert torch.equal(t.view(torch.int32), ops.split_pack(ops.weight_transpose_scale(w, s), 2048.0).data.view(torch.int32))
    for (N, C) in ((64, 64), (512, 512), (32, 96)):
        w = (torch.randn(N, C, 3, 3, generator=g) * 0.05).cuda()
        s = (torch.rand(N, generator=g) + 0.5).cuda()
        outs = {k: torch.empty(ops.prep_shape(k, w), device="cuda") for k in ("wino", "col", "uflip", "flip9")}
        ops.res5_weight_prep([("wino", w, None, outs["wino"], 1024.0), ("col", w, None, outs["col"], 4096.0),
                              ("uflip", w, s, outs["uflip"], 512.0), ("flip9", w, s, outs["flip9"], 2048.0)])
        wflip = ops.conv3x3_weight_flip(w, s)
        want = {"wino": ops.split_pack(ops.winograd_pack_weight(w), 1024.0), "col": ops.split_pack(ops.pack_conv3x3_weight(w), 4096.0),
                "uflip": ops.split_pack(ops.winograd_pack_weight(wflip), 512.0), "flip9": ops.split_pack(ops.pack_conv3x3_weight(wflip), 2048.0)}
        for k in outs:
            assert torch.equal(outs[k].view(torch.int32), want[k].data.view(torch.int32)), (N, C, k)
    # the range guard: a scale that no longer covers the weight
    ops.split_overflow_reset("cuda")
    w = torch.full((32, 32), 3.0, device="cuda")
    ops.res5_weight_prep([("plain", w, None, torch.empty(32, 32, device="cuda"), 2.0 ** 20)])
    assert ops.split_overflow_raised("cuda")
    ops.split_overflow_reset("cuda")


def test_training_steps_with_prepared_operands_equal_the_first_step(pkg, oracle):
    """The first training step of a stage chooses the operand scales (multi-launch chain, host reads); the following ones build
    every operand in one launch (Res5Stage.train_operands).  Same weights, same input -> the same outputs and gradients, bit
    for bit, and the one-launch path is really taken."""
    from locov_amd import res5_train
    R, (in_ch, mid, out_ch) = 21, (128, 64, 256)
    res5, params = _stage(pkg, oracle, in_ch, mid, out_ch, seed=8)
    gen = torch.Generator().manual_seed(29)
    N, H, W = 2, 26, 43
    feat = torch.randn(N, in_ch, H, W, generator=gen).cuda()
    wh = torch.rand(R, 2, generator=gen) * 300 + 30
    xy = torch.rand(R, 2, generator=gen) * 300
    rois = torch.cat([torch.randint(0, N, (R, 1), generator=gen).float(), xy, xy + wh], dim=1).cuda()
    named = dict(res5.named_parameters())
    keys = _weight_keys(params)

    def step():
        res5.zero_grad()
        f = feat.clone().requires_grad_(True)
        nhwc = res5_train.to_nhwc(f)
        st = res5_train.Res5Step(res5, True, f.device, N * 13 * 22 + 49 * R)
        rows = res5_train.grid_segment(st, nhwc)
        x0 = res5_train.roi_segment(st, nhwc, rois, 14, 1.0 / 16, 0, True)
        grid, box = st.outputs([rows, x0], [False, True])
        (grid.square().mean() + box.sum()).backward()
        torch.cuda.synchronize()
        return len(st.operands.ready), grid.detach().clone(), box.detach().clone(), f.grad.clone(), {k: named[k].grad.clone() for k in keys}

    first = step()
    assert first[0] == 0                                         # cold: scales chosen on the way
    with torch.no_grad():
        for k in keys:
            named[k].add_(0.0)                                   # an "optimizer step": every weight's version moves
    second = step()
    assert second[0] == 2 * 7 + 4 * 3                            # warm: 26 operands from one launch
    assert torch.equal(first[1], second[1]) and torch.equal(first[2], second[2])
    assert rel_err(second[3], first[3]) < 1e-6                   # (the map's gradient: ROIAlign's backward adds with fp32 atomics)
    for k in keys:
        assert torch.equal(first[4][k], second[4][k]), k


def test_amax_bound_and_the_scale_it_implies(pkg):
    """ops.amax_bound: max_i (mul_i * max |x_i|) of small tensors in a lazy operand-scale slot -- the scale a split GEMM derives from
    it puts the bounded tensor's max in [2^12, 2^13); folding into an existing slot keeps the larger value."""
    ops = pkg.ops
    g = torch.Generator().manual_seed(5)
    a = (torch.randn(800, 2048, generator=g) * 3e-3).cuda()
    b = (torch.randn(4200, 2048, generator=g) * 1e-5).cuda()
    slot = ops.amax_bound([a, b], [1.0 / 49, 1.0])
    torch.cuda.synchronize()
    want = max(float(a.abs().max()) / 49, float(b.abs().max()))
    got = float(slot.view(torch.int32)[2:3].view(torch.float32))
    assert abs(got - want) <= 1e-6 * want and float(slot[0]) == 0.0            # a LAZY slot: consumers derive the scale
    x = torch.randn(256, 64, generator=g).cuda()
    x = x / x.abs().max() * want                                               # a tensor whose max is the bound
    w = ops.split_pack((torch.randn(32, 64, generator=g) * 0.1).cuda())
    ops.split_overflow_reset("cuda")
    y = ops.linear_split_ex(x, w, x_scale_dev=slot)
    assert not ops.split_overflow_raised("cuda")
    assert rel_err(y, x.double() @ ops.split_unpack(w.data, w.scale).double().t()) < 3e-6
    ops.amax_bound([a], [1e-3], slot=slot)                                     # a smaller contribution: unchanged
    torch.cuda.synchronize()
    assert float(slot.view(torch.int32)[2:3].view(torch.float32)) == got
    ops.amax_bound([a], [10.0], slot=slot)                                     # a larger one: raised
    torch.cuda.synchronize()
    assert float(slot.view(torch.int32)[2:3].view(torch.float32)) > got
    with pytest.raises(ValueError):
        ops.amax_bound([torch.zeros(6, device="cuda")], [1.0])


def test_res5_step_edge_cases(pkg, oracle):
    """Res5Step: an empty segment (an image batch without sampled proposals) between two live ones, spare capacity, the capacity
    check, and three segments forwarded together -- outputs equal the single-segment calls, gradients flow to every input."""
    from locov_amd import res5_train
    in_ch, mid, out_ch = 128, 64, 256
    res5, params = _stage(pkg, oracle, in_ch, mid, out_ch, seed=19)
    gen = torch.Generator().manual_seed(3)
    xa = torch.randn(5 * 49, in_ch, generator=gen).cuda().requires_grad_(True)            # 5 proposals
    xg = torch.randn(1 * 9 * 11, in_ch, generator=gen).cuda().requires_grad_(True)        # one 9 x 11 grid
    xe = torch.zeros(0, in_ch, device="cuda", requires_grad=True)
    want_a = res5_train.res5_rows(res5, xa.detach().clone().requires_grad_(True), 5, 7, 7, pooled=True, split=True)
    want_g = res5_train.res5_rows(res5, xg.detach().clone().requires_grad_(True), 1, 9, 11, pooled=False, split=True)
    step = res5_train.Res5Step(res5, True, xa.device, 5 * 49 + 99 + 500)
    for x in (xa, xe, xg):
        step.input_rows(x.shape[0]).copy_(x.detach())
    with pytest.raises(ValueError, match="exceed the capacity"):
        step.input_rows(10 ** 6)
    segs = step.forward_segments([(5, 7, 7), (0, 7, 7), (1, 9, 11)])
    assert [s.rows for s in segs] == [245, 0, 99] and step.filled == 344
    out_a, out_e, out_g = step.outputs([xa, xe, xg], [True, True, False])
    assert tuple(out_e.shape) == (0, out_ch) and torch.equal(out_a, want_a) and torch.equal(out_g, want_g)
    (out_a.sum() + out_g.square().mean() + out_e.sum()).backward()
    assert xa.grad is not None and float(xa.grad.abs().max()) > 0 and float(xg.grad.abs().max()) > 0
    assert xe.grad is None or xe.grad.numel() == 0
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for k, p in res5.named_parameters() if k.endswith(".weight") and ".norm." not in k)


# ------------------------------------------------------------------------------------------------ the heads
def _train_heads(pkg, oracle, backend, dtype, small=True):
    from locov_amd.structures import ShapeSpec
    cfg = pkg.config.get_cfg()
    if small:
        cfg.MODEL.RESNETS.RES2_OUT_CHANNELS = 32        # res5: 128 -> (64) -> 256
        cfg.MODEL.RESNETS.WIDTH_PER_GROUP = 8
        cfg.MODEL.ROI_BOX_HEAD.EMB_DIM = 96
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    cfg.MODEL.ROI_BOX_HEAD.FREEZE_EMB_PRED = False
    cfg.MODEL.ROI_HEADS.NAME = "EmbeddingProposalsRes5ROIHeads"
    cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 24 if small else 200
    cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION = 1.0
    cfg.MODEL.ROI_HEADS.DETACH_CLASS_PREDICTOR = True
    cfg.MODEL.ROI_BOX_HEAD.RES5_BACKEND = backend
    cfg.MODEL.ROI_BOX_HEAD.RES5_DTYPE = dtype
    c_in = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS * 4
    torch.manual_seed(3)
    heads = pkg.build_roi_heads(cfg, {"res4": ShapeSpec(channels=c_in, stride=16)})
    params = oracle.make_res5_params(9, in_ch=c_in, mid=cfg.MODEL.RESNETS.WIDTH_PER_GROUP * 8, out_ch=heads.output_shape)
    heads.res5.load_state_dict(params)
    rng = np.random.default_rng(9)
    h = oracle.synth_head(rng, heads.output_shape, cfg.MODEL.ROI_BOX_HEAD.EMB_DIM, 80)
    heads = heads.cuda().train()
    heads.box_predictor.set_class_embeddings(h["cls_w"])
    heads.num_classes = heads.box_predictor.num_classes
    return heads, c_in


def _train_batch(pkg, oracle, n_img, r, n_gt, seed):
    from locov_amd.structures import Boxes, Instances
    rng = np.random.default_rng(seed)
    props, targets = [], []
    for _ in range(n_img):
        gt = oracle.synth_boxes(rng, n_gt)
        b = oracle.synth_boxes(rng, r)
        b[:n_gt] = gt + rng.uniform(-4, 4, gt.shape).astype(np.float32)      # some proposals overlap the ground truth
        b[:, 2:] = np.maximum(b[:, 2:], b[:, :2] + 1.0)
        p = Instances((800, 1333))
        p.proposal_boxes = Boxes(torch.from_numpy(b).cuda())
        p.objectness_logits = torch.zeros(r, device="cuda")
        t = Instances((800, 1333))
        t.gt_boxes = Boxes(torch.from_numpy(gt).cuda())
        t.gt_classes = torch.from_numpy(rng.integers(0, 80, n_gt)).cuda()
        props.append(p)
        targets.append(t)
    return props, targets


@pytest.mark.parametrize("dtype", ["fp32", "f16x2"])
def test_training_forward_backward_matches_the_stock_library_path(pkg, oracle, dtype):
    """EmbeddingProposalsRes5ROIHeads.forward with targets (roi_emb_heads.py:311-349), one training step's forward and
    backward: the hand-written path (RES5_BACKEND hip) against the same module on torch conv2d / MIOpen autograd
    (RES5_BACKEND miopen) -- same sampled proposals (same RNG seed), losses, region features, grid features and the
    gradients of the res4 map and of every trainable parameter."""
    outs = {}
    for backend in ("miopen", "hip"):
        heads, c_in = _train_heads(pkg, oracle, backend, dtype)
        feat = torch.randn(2, c_in, 50, 84, generator=torch.Generator().manual_seed(5)).cuda().requires_grad_(True)
        props, targets = _train_batch(pkg, oracle, 2, 60, 5, seed=31)
        torch.manual_seed(77)                                  # the proposal sampler draws from the global RNG
        grid, box_feats, sampled, losses = heads(None, {"res4": feat}, props, targets)
        loss = losses["loss_box_reg"] + losses["loss_cls"] + 1e-3 * grid.square().mean() + 1e-2 * torch.cat(box_feats).square().mean()
        loss.backward()
        outs[backend] = (grid.detach(), torch.cat(box_feats).detach(), float(losses["loss_box_reg"]), feat.grad.clone(),
                         {k: p.grad.clone() for k, p in heads.named_parameters() if p.grad is not None})
    gm, bm, lm, fm, pm = outs["miopen"]
    gh, bh, lh, fh, ph = outs["hip"]
    assert tuple(gh.shape) == tuple(gm.shape) == (2, 256, 25, 42)
    tol = 2e-4            # two fp32 evaluations against each other (neither is the float64 reference)
    assert rel_err(gh, gm) < tol and rel_err(bh, bm) < tol and abs(lh - lm) <= tol * max(abs(lm), 1e-3)
    # gradients: two fp32 forwards disagree on a handful of ReLU ties (see _float64_stage), each of which moves single
    # gradient entries by ~1e-3 of the largest one; the exact gates are the float64 tests above, here the bulk must agree
    rel_l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    assert rel_l2(fh, fm) < 5e-3 and rel_err(fh, fm) < 5e-2
    assert set(ph) == set(pm) and any(k.startswith("res5.") for k in ph)
    worst = max((rel_l2(ph[k], pm[k]), k) for k in ph)
    assert worst[0] < 5e-3, worst


def _train_batch_sized(pkg, oracle, sizes, r, n_gts, seed):
    """_train_batch with every image its own (h, w), its own number of proposals (r: int or list) and ground-truth count."""
    from locov_amd.structures import Boxes, Instances
    rng = np.random.default_rng(seed)
    props, targets = [], []
    rs = r if isinstance(r, (list, tuple)) else [r] * len(sizes)
    for (h, w), ri, n_gt in zip(sizes, rs, n_gts):
        scale = np.array([w / 1333.0, h / 800.0, w / 1333.0, h / 800.0], dtype=np.float32)
        gt = oracle.synth_boxes(rng, n_gt) * scale
        b = oracle.synth_boxes(rng, ri) * scale
        if n_gt:
            b[:n_gt] = gt + rng.uniform(-4, 4, gt.shape).astype(np.float32)
        b = np.maximum(b, 0)
        b[:, 2:] = np.maximum(b[:, 2:], b[:, :2] + 1.0)
        p = Instances((h, w))
        p.proposal_boxes = Boxes(torch.from_numpy(b.astype(np.float32)).cuda())
        p.objectness_logits = torch.zeros(ri, device="cuda")
        t = Instances((h, w))
        t.gt_boxes = Boxes(torch.from_numpy(gt.astype(np.float32).reshape(-1, 4)).cuda())
        t.gt_classes = torch.from_numpy(rng.integers(0, 80, n_gt)).cuda()
        props.append(p)
        targets.append(t)
    return props, targets


def test_changing_map_sizes_in_one_process_match_the_stock_library_path(pkg, oracle):
    """the reference trains multi-scale (configs/coco_stt.yaml:54 MIN_SIZE_TRAIN (640 ... 800), batches padded to
    their largest image), so consecutive steps see DIFFERENT res4 maps, proposal counts and ground-truth counts.  Three steps
    with three map sizes through ONE pair of modules (hand-written path / torch conv2d autograd), each followed by an SGD step:
    every step's outputs and gradients agree -- the packed operands, workspaces, cached index tensors and the Res5Step matrices
    must follow the shape, not remember the previous one.  The second batch holds an image WITHOUT ground truth, the third an
    image with fewer proposals than the sampling budget (the forward cannot speculate and waits for the true counts)."""
    heads = {b: _train_heads(pkg, oracle, b, "f16x2")[0] for b in ("miopen", "hip")}
    c_in = 128
    opts = {b: torch.optim.SGD([p for p in h.parameters() if p.requires_grad], lr=1e-3) for b, h in heads.items()}
    cases = [  # (map H, W), per-image sizes, proposals per image, GT per image
        ((50, 84), [(800, 1333), (800, 1200)], 60, [5, 3]),
        ((40, 60), [(640, 960), (640, 853)], [48, 70], [0, 9]),
        ((67, 67), [(1067, 800), (800, 1067)], [90, 16], [15, 2]),
        ((50, 84), [(800, 1333), (800, 1200)], 60, [5, 3]),          # ... and back to the first shape
    ]
    rel_l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    for it, ((mh, mw), sizes, r, n_gts) in enumerate(cases):
        outs = {}
        for backend, h in heads.items():
            feat = torch.randn(2, c_in, mh, mw, generator=torch.Generator().manual_seed(50 + it)).cuda().requires_grad_(True)
            props, targets = _train_batch_sized(pkg, oracle, sizes, r, n_gts, seed=60 + it)
            opts[backend].zero_grad(set_to_none=True)
            torch.manual_seed(90 + it)
            grid, box_feats, sampled, losses = h(None, {"res4": feat}, props, targets)
            loss = losses["loss_box_reg"] + losses["loss_cls"] + 1e-3 * grid.square().mean() + 1e-2 * torch.cat(box_feats).square().mean()
            loss.backward()
            outs[backend] = (grid.detach(), torch.cat(box_feats).detach(), float(losses["loss_box_reg"]), feat.grad.clone(),
                             {k: p.grad.clone() for k, p in h.named_parameters() if p.grad is not None}, [len(x) for x in sampled])
            opts[backend].step()
        gm, bm, lm, fm, pm, nm = outs["miopen"]
        gh, bh, lh, fh, ph, nh = outs["hip"]
        assert tuple(gh.shape) == tuple(gm.shape) == (2, 256, (mh + 1) // 2, (mw + 1) // 2), (it, gh.shape)
        assert nh == nm, (it, nh, nm)
        tol = 3e-4            # two fp32 evaluations against each other, on weights that took the previous steps' (slightly different) updates
        assert rel_err(gh, gm) < tol and rel_err(bh, bm) < tol and abs(lh - lm) <= tol * max(abs(lm), 1e-3), (it, rel_err(gh, gm), rel_err(bh, bm), lh, lm)
        assert rel_l2(fh, fm) < 5e-3, (it, rel_l2(fh, fm))
        assert set(ph) == set(pm)
        worst = max((rel_l2(ph[k], pm[k]), k) for k in ph)
        assert worst[0] < 5e-3, (it, worst)
    st = heads["hip"].stats
    assert st["forwards"] == 4 and st.get("unspeculated", 0) == 1 and st.get("speculated", 0) == 3, st      # (16 proposals < the budget of 24)


def test_operand_scales_are_refreshed_without_a_host_wait(pkg, oracle, monkeypatch):
    """Every REFRESH steps the remembered operand scales are chosen again from max |w| / max |s| sent to pinned memory behind an event
    (TrainOperands._start_refresh / _adopt_refresh): the one-launch preparation is never left (the synchronous form sends every
    REFRESH-th step to the host-read chain), the new scales cover their operands with the chain's headroom or one power of two
    less, and the training run stays what it is with scales that are never refreshed."""
    from locov_amd import res5 as res5_mod
    ops = pkg.ops
    w = torch.randn(64, 32, 3, 3, generator=torch.Generator().manual_seed(1)).cuda()
    assert float(ops.winograd_pack_weight(w).abs().max()) <= res5_mod.TrainOperands.WINO_GAIN * float(w.abs().max())
    runs = {}
    for refresh in (2, 10 ** 9):
        monkeypatch.setattr(res5_mod.TrainOperands, "REFRESH", refresh)
        heads, c_in = _train_heads(pkg, oracle, "hip", "f16x2")
        opt = torch.optim.SGD([p for p in heads.parameters() if p.requires_grad], lr=1e-3)
        feat = torch.randn(2, c_in, 50, 84, generator=torch.Generator().manual_seed(5)).cuda().requires_grad_(True)
        props, targets = _train_batch(pkg, oracle, 2, 60, 5, seed=31)
        losses, ready, adopted = [], [], 0
        for it in range(12):
            opt.zero_grad(set_to_none=True)
            torch.manual_seed(77 + it)
            before = dict(heads.res5._scales)
            grid, box_feats, sampled, ls = heads(None, {"res4": feat}, props, targets)
            ready.append(len(heads.res5.__dict__["_train_ops"][1].ready))
            adopted += any(v[1] < before.get(k, (0, -1))[1] for k, v in heads.res5._scales.items())
            loss = ls["loss_box_reg"] + ls["loss_cls"] + 1e-3 * grid.square().mean() + 1e-2 * torch.cat(box_feats).square().mean()
            loss.backward()
            losses.append(float(loss.detach()))
            opt.step()
        runs[refresh] = (losses, ready, adopted)
        # every remembered scale covers its operand: |scale * operand| < 2^13 (the chain's target), and not by more than 2^3 below it
        stage = heads.res5
        for blk in stage:
            for conv in (blk.conv1, blk.conv3):
                rec = stage._scales[(id(conv), False)]
                top = rec[0] * float(conv.weight.abs().max())
                assert 2.0 ** 9 <= top < 2.0 ** 13, top
    assert runs[2][1] == [0] + [26] * 11 and runs[10 ** 9][1] == [0] + [26] * 11
    assert runs[2][2] >= 2 and runs[10 ** 9][2] == 0                     # scales re-chosen (two steps behind each request) / never
    for a, b in zip(runs[2][0], runs[10 ** 9][0]):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(b)), (a, b)


def test_one_launch_operands_with_a_frozen_backbone(pkg, oracle):
    """A res4 map that needs no gradient (frozen backbone): block 0's data-gradient operands are never asked for, so the on-demand
    chain never learns their scales -- the one-launch preparation must still take over from the second step (it learns the
    missing scales itself), and the steps' losses and Res5 weight gradients equal those of the on-demand chain bit for bit."""
    from locov_amd import res5 as res5_mod
    runs = {}
    for one_launch in (False, True):
        was, res5_mod._ONE_LAUNCH_PREP = res5_mod._ONE_LAUNCH_PREP, one_launch
        try:
            heads, c_in = _train_heads(pkg, oracle, "hip", "f16x2")
            opt = torch.optim.SGD([p for p in heads.parameters() if p.requires_grad], lr=1e-3)
            feat = torch.randn(2, c_in, 50, 84, generator=torch.Generator().manual_seed(5)).cuda()        # requires no gradient
            props, targets = _train_batch(pkg, oracle, 2, 60, 5, seed=31)
            seen, ready = [], []
            for it in range(3):
                opt.zero_grad(set_to_none=True)
                torch.manual_seed(77 + it)
                grid, box_feats, sampled, losses = heads(None, {"res4": feat}, props, targets)
                ready.append(len(heads.res5.__dict__["_train_ops"][1].ready))
                loss = losses["loss_box_reg"] + losses["loss_cls"] + 1e-3 * grid.square().mean() + 1e-2 * torch.cat(box_feats).square().mean()
                loss.backward()
                seen.append((float(loss.detach()), {k: p.grad.clone() for k, p in heads.named_parameters()
                                                    if p.grad is not None and k.startswith("res5.")}))
                opt.step()
            torch.cuda.synchronize()
        finally:
            res5_mod._ONE_LAUNCH_PREP = was
        runs[one_launch] = (seen, ready)
    assert runs[False][1] == [0, 0, 0] and runs[True][1] == [0, 26, 26]
    for (la, ga), (lb, gb) in zip(runs[False][0], runs[True][0]):
        assert la == lb and set(ga) == set(gb) and len(ga) >= 10
        for k in ga:
            assert torch.equal(ga[k], gb[k]), k


@pytest.mark.parametrize("mode", ["deferred", "sync"])
def test_training_forward_out_of_range_never_reaches_the_losses_or_the_parameters(pkg, oracle, mode, monkeypatch):
    """A training forward whose activations leave the split arithmetic's range (|x| >= 4094).  RES5_TRAIN_GUARD "sync"
    (default): the forward is repeated on the f32 MFMA inside the step -- the reference's values.  "deferred": no host read
    inside the step -- the forward's outputs, the ROI heads' own losses and every gradient of the step's ROI-head path
    (Res5 AND the predictor: a skipped step) are zero on the device, also when a LATER forward's labelling read has cleared
    the guard before this forward's backward runs (gradient accumulation); the next step's labelling read reports it (one
    RuntimeWarning) and the module runs on the f32 MFMA from then on, where the same input gives the f32 path's results."""
    import warnings
    monkeypatch.setenv("LOCOV_RES5_TRAIN_GUARD", mode)
    heads, c_in = _train_heads(pkg, oracle, "hip", "f16x2")
    assert heads.res5_train_guard == mode
    monkeypatch.delenv("LOCOV_RES5_TRAIN_GUARD")
    assert _train_heads(pkg, oracle, "hip", "f16x2")[0].res5_train_guard == "sync"          # the default keeps the reference's values
    ref, _ = _train_heads(pkg, oracle, "hip", "fp32")
    gen = torch.Generator().manual_seed(6)
    base = torch.randn(2, c_in, 50, 84, generator=gen).cuda()
    props, targets = _train_batch(pkg, oracle, 2, 60, 5, seed=32)

    def step(h, scale, seed):
        h.zero_grad()
        feat = (base * scale).requires_grad_(True)
        torch.manual_seed(seed)
        grid, box_feats, sampled, losses = h(None, {"res4": feat}, props, targets)
        (sum(losses.values()) + grid.mean() + sum(b.sum() for b in box_feats) * 1e-3).backward()
        torch.cuda.synchronize()
        return grid, box_feats, losses, feat.grad, {k: p.grad for k, p in h.named_parameters() if p.grad is not None}

    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        step(heads, 1.0, 1)                                              # in range: nothing is reported, nothing zeroed
        assert heads.res5_dtype == "f16x2"
    big = 3.0e4                                                          # a res4 map of ~1e5: far outside fp16 at the activation scale
    want = step(ref, big, 2)
    if mode == "sync":
        with pytest.warns(RuntimeWarning, match="repeated"):
            got = step(heads, big, 2)
        assert rel_err(got[0], want[0]) < 1e-6 and rel_err(got[3], want[3]) < 1e-5
        return
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)                   # deferred: silent inside the step ...
        grid, box_feats, losses, gfeat, gparams = step(heads, big, 2)
    assert float(grid.abs().max()) == 0.0 and all(float(b.abs().max()) == 0.0 for b in box_feats)
    assert all(bool(torch.isfinite(v).all()) for v in losses.values())
    assert bool(torch.isfinite(gfeat).all()) and float(gfeat.abs().max()) == 0.0
    assert all(float(v) == 0.0 for v in losses.values())                 # the module's own losses are those of a skipped step
    for k, g in gparams.items():                                         # ... and so is every gradient they send back
        assert bool(torch.isfinite(g).all()) and float(g.abs().max()) == 0.0, k
    # gradient accumulation: forward A (out of range), forward B (whose labelling read finds and CLEARS the guard), then A's
    # backward -- A's own copy of the word still zeroes its gradients
    monkeypatch.setenv("LOCOV_RES5_TRAIN_GUARD", "deferred")
    other, _ = _train_heads(pkg, oracle, "hip", "f16x2")
    other.zero_grad()
    feat_a = (base * big).requires_grad_(True)
    torch.manual_seed(2)
    grid_a, box_a, _, losses_a = other(None, {"res4": feat_a}, props, targets)
    with pytest.warns(RuntimeWarning, match="ZEROED on the device"):
        torch.manual_seed(3)
        other(None, {"res4": base.clone().requires_grad_(True)}, props, targets)
    (sum(losses_a.values()) + grid_a.mean() + sum(b.sum() for b in box_a) * 1e-3).backward()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(feat_a.grad).all()) and float(feat_a.grad.abs().max()) == 0.0
    for k, p in other.named_parameters():
        if p.grad is not None:
            assert bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) == 0.0, k
    with pytest.warns(RuntimeWarning, match="ZEROED on the device"):     # ... and reported by the next step's one host read
        got = step(heads, big, 2)
    assert heads.res5_dtype == "fp32"
    assert rel_err(got[0], want[0]) < 1e-6 and rel_err(got[3], want[3]) < 1e-5
    for k in want[4]:
        if k.startswith("res5."):
            assert rel_err(got[4][k], want[4][k]) < 1e-5, k


def _ignore_band_heads(pkg, oracle, dtype):
    """_train_heads with a sampler that has an IGNORE band (Matcher [0.3, 0.7] -> [0, -1, 1]): the one configuration in which a
    speculated sample can miss -- an image with enough proposals whose CANDIDATES (labels != -1) do not fill the budget."""
    from locov_amd.roi_heads.roi_emb_heads import Matcher
    heads, c_in = _train_heads(pkg, oracle, "hip", dtype)
    heads.proposal_matcher = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=False)
    return heads, c_in


def _miss_batch(n_img, r, budget, seed):
    from locov_amd.structures import Boxes, Instances
    from tests import ddp_worker
    props, targets = [], []
    for i in range(n_img):
        b, gt = ddp_worker.ignore_band_boxes(r, budget, seed * 10 + i)
        p = Instances((800, 1333))
        p.proposal_boxes = Boxes(torch.from_numpy(b).cuda())
        p.objectness_logits = torch.zeros(len(b), device="cuda")
        t = Instances((800, 1333))
        t.gt_boxes = Boxes(torch.from_numpy(gt).cuda())
        t.gt_classes = torch.tensor([3, 17, 42], device="cuda")
        props.append(p)
        targets.append(t)
    return props, targets


def test_speculation_miss_and_range_trip_in_one_step(pkg, oracle, monkeypatch):
    """VERDICT r5 item 4 -- the training forward's retry machine through TWO legs in one step: the speculated sample misses
    (repeat from the true counts) AND that repeated forward leaves the split arithmetic's range (RES5_TRAIN_GUARD "sync": repeat on
    the f32 MFMA): attempt() runs three times, the step's outputs, losses and gradients are those of the fp32 module on the same
    draw, the counters say what happened, and roi_head/num_{fg,bg}_samples is logged ONCE."""
    import warnings
    from locov_amd.roi_heads import roi_emb_heads as mod
    heads, c_in = _ignore_band_heads(pkg, oracle, "f16x2")
    ref, _ = _ignore_band_heads(pkg, oracle, "fp32")
    assert heads.res5_train_guard == "sync"
    base = torch.randn(2, c_in, 50, 84, generator=torch.Generator().manual_seed(6)).cuda()
    props, targets = _miss_batch(2, 60, heads.batch_size_per_image, seed=3)
    puts = []
    monkeypatch.setattr(mod._EVENTS, "put_scalar", lambda name, value: puts.append(name))

    def step(h, scale, seed):
        h.zero_grad()
        feat = (base * scale).requires_grad_(True)
        torch.manual_seed(seed)
        grid, box_feats, sampled, losses = h(None, {"res4": feat}, props, targets)
        (sum(losses.values()) + grid.mean() + sum(b.sum() for b in box_feats) * 1e-3).backward()
        torch.cuda.synchronize()
        return grid, torch.cat(box_feats), losses, feat.grad, {k: p.grad for k, p in h.named_parameters() if p.grad is not None}, [len(x) for x in sampled]

    big = 3.0e4
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        want = step(ref, big, 2)
    assert ref.stats == {"forwards": 1, "speculated": 1, "speculation_misses": 1}, ref.stats
    assert all(n == heads.batch_size_per_image // 3 + 3 for n in want[5]), want[5]        # the under-filled sample: bg candidates + the GT
    del puts[:]
    with pytest.warns(RuntimeWarning, match="repeated"):
        got = step(heads, big, 2)
    assert heads.stats == {"forwards": 1, "speculated": 1, "speculation_misses": 1, "guard_trips": 1, "fp32_repeats": 1}, heads.stats
    assert sorted(puts) == ["roi_head/num_bg_samples", "roi_head/num_fg_samples"], puts
    assert got[5] == want[5]
    assert rel_err(got[0], want[0]) < 1e-6 and rel_err(got[1], want[1]) < 1e-6 and rel_err(got[3], want[3]) < 1e-5
    for k in want[2]:
        assert abs(float(got[2][k]) - float(want[2][k])) <= 1e-6 * max(1.0, abs(float(want[2][k]))), k
    assert set(got[4]) == set(want[4])
    for k in want[4]:
        assert rel_err(got[4][k], want[4][k]) < 1e-5, k
    # the next (in-range, budget-filling) step is back on the split arithmetic and on the speculated path, silently
    props2, targets2 = _train_batch(pkg, oracle, 2, 60, 5, seed=32)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        heads.zero_grad()
        torch.manual_seed(5)
        heads(None, {"res4": base.clone().requires_grad_(True)}, props2, targets2)
    assert heads.res5_dtype == "f16x2" and heads.stats["speculated"] == 2 and heads.stats["speculation_misses"] == 1


@pytest.mark.parametrize("order", ["miss_then_trip", "trip_then_miss"])
def test_retry_legs_under_gradient_accumulation(pkg, oracle, order):
    """Two forwards, then two backwards (gradient accumulation): one forward's speculated sample misses, the other leaves the split
    arithmetic's range and is repeated on the f32 MFMA (which drops the remembered operand scales while the FIRST forward's graph,
    with its own operand set, still waits for its backward).  The accumulated gradients equal those of the same two steps run one
    after the other (forward, backward, forward, backward)."""
    import warnings
    heads, c_in = _ignore_band_heads(pkg, oracle, "f16x2")
    base = torch.randn(2, c_in, 50, 84, generator=torch.Generator().manual_seed(8)).cuda()
    miss = _miss_batch(2, 60, heads.batch_size_per_image, seed=4)
    plain = _train_batch(pkg, oracle, 2, 60, 5, seed=33)
    legs = [(miss, 1.0, 11), (plain, 3.0e4, 12)]
    if order == "trip_then_miss":
        legs = legs[::-1]
    opt = torch.optim.SGD([p for p in heads.parameters() if p.requires_grad], lr=1e-4)
    for it in range(2):                                  # (two plain steps first: the operand scales are remembered, the one-launch
        opt.zero_grad(set_to_none=True)                  #  preparation is what the steps under test run on)
        torch.manual_seed(20 + it)
        g, bf, _, ls = heads(None, {"res4": base.clone().requires_grad_(True)}, *plain)
        (sum(ls.values()) + g.mean()).backward()
        opt.step()

    def forward(leg):
        (props, targets), scale, seed = leg
        feat = (base * scale).requires_grad_(True)
        torch.manual_seed(seed)
        grid, box_feats, sampled, losses = heads(None, {"res4": feat}, props, targets)
        return feat, sum(losses.values()) + grid.mean() * 1e-3 + sum(b.sum() for b in box_feats) * 1e-6

    def grads(feats):
        return [f.grad.clone() for f in feats], {k: p.grad.clone() for k, p in heads.named_parameters() if p.grad is not None}

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        heads.zero_grad()
        feats = []
        for leg in legs:                                 # one after the other
            f, loss = forward(leg)
            loss.backward()
            feats.append(f)
        torch.cuda.synchronize()
        want_f, want_p = grads(feats)
        heads.stats.clear()
        heads.zero_grad()
        pend = [forward(leg) for leg in legs]            # both forwards ...
        for _, loss in pend:                             # ... then both backwards
            loss.backward()
        torch.cuda.synchronize()
        got_f, got_p = grads([f for f, _ in pend])
    assert heads.stats.get("speculation_misses") == 1 and heads.stats.get("fp32_repeats") == 1 and heads.stats["forwards"] == 2, heads.stats
    for a, b in zip(got_f, want_f):
        assert bool(torch.isfinite(a).all()) and rel_err(a, b) < 1e-5
    assert set(got_p) == set(want_p) and any(k.startswith("res5.") for k in got_p)
    for k in want_p:
        assert bool(torch.isfinite(got_p[k]).all()) and rel_err(got_p[k], want_p[k]) < 1e-5, k


def test_training_step_at_config_sizes_runs_on_the_hip_kernels(pkg, oracle):
    """configs/coco_lsm.yaml sizes (Res5 1024 -> 512 -> 2048, 200 sampled proposals per image): one forward + backward
    on the hand-written path; finite gradients for every Res5 convolution and the res4 map."""
    heads, c_in = _train_heads(pkg, oracle, "hip", "f16x2", small=False)
    feat = torch.randn(2, c_in, 50, 84, generator=torch.Generator().manual_seed(5)).cuda().requires_grad_(True)
    props, targets = _train_batch(pkg, oracle, 2, 300, 8, seed=41)
    grid, box_feats, sampled, losses = heads(None, {"res4": feat}, props, targets)
    assert tuple(grid.shape) == (2, 2048, 25, 42) and sum(len(s) for s in sampled) == 400
    (losses["loss_box_reg"] + grid.mean() + torch.cat(box_feats).mean()).backward()
    assert torch.isfinite(feat.grad).all() and float(feat.grad.abs().max()) > 0
    n = 0
    for k, p in heads.res5.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0, k
        n += 1
    assert n == 10


# ------------------------------------------------------------------------------------------------ split-arithmetic backward
def test_split_scale_from_amax(pkg):
    ops = pkg.ops
    g = torch.Generator().manual_seed(2)
    for amax in (3.7e-6, 0.9, 1.0, 517.25):
        x = torch.randn(1000, 64, generator=g).cuda()
        x = x / x.abs().max() * amax
        sc = ops.split_scale_from_amax(x).cpu()
        s = float(sc[0])
        assert s == 2.0 ** round(np.log2(s)) and float(sc[1]) == 1.0 / s              # a power of two and its inverse
        assert 2.0 ** 12 <= float(x.abs().max()) * s < 2.0 ** 13
    assert float(ops.split_scale_from_amax(torch.zeros(16, 8, device="cuda"))[0]) == 1.0


@pytest.mark.parametrize("M,N,K", [(1000, 128, 128), (777, 64, 96), (4097, 256, 36), (39200, 512, 256), (31, 4, 2048)])
def test_gemm_tn_split_vs_float64(pkg, M, N, K):
    """Weight-gradient GEMM in split arithmetic: gradient-sized entries (1e-5) against activations, float64 reference; error
    of the order of the f32-MFMA TN kernel's."""
    ops = pkg.ops
    g = torch.Generator().manual_seed(M + K)
    a = (torch.randn(M, N, generator=g) * 1e-5).cuda()
    a[::7] *= 30.0                                                        # a spread of magnitudes inside one tensor
    b = torch.relu(torch.randn(M, K, generator=g)).cuda() * 3.0
    s = (torch.rand(N, generator=g) + 0.5).cuda()
    want = (a.double().t() @ b.double()) * s.double()[:, None]
    sc = ops.split_scale_from_amax(a)
    ops.split_overflow_reset(a.device)
    got = ops.gemm_tn_split(a, b, s, sc, 16.0)
    assert not ops.split_overflow_raised(a.device)
    err, err32 = rel_err(got, want), rel_err(ops.gemm_tn(a, b, s), want)
    assert err < 3e-6 and err < 4 * err32 + 1e-6, (err, err32)
    assert torch.equal(got, ops.gemm_tn_split(a, b, s, sc, 16.0))
    b[5, 3] = 5000.0                                                      # 16 * 5000 >= 65504: the guard sees it
    ops.gemm_tn_split(a, b, s, sc, 16.0)
    assert ops.split_overflow_raised(a.device)
    ops.split_overflow_reset(a.device)


@pytest.mark.parametrize("M,N,K,scale", [(1000, 128, 128, 16.0), (777, 64, 96, 0.25), (4097, 256, 160, 16.0), (39200, 512, 256, 0.25), (31, 4, 2048, 16.0)])
def test_gemm_tn_split_with_a_presplit_operand_is_the_same_bits(pkg, M, N, K, scale):
    """locov_gemm_tn_f32_split_b: the activation operand handed over in the split layout (what a producer's epilogue or the forward's
    Winograd input transform wrote) is transposed on its way into LDS instead of converted -- the result equals the converting
    launch's bit for bit, for ragged M chunks, N / K below and above a tile, and both operand scales."""
    ops = pkg.ops
    g = torch.Generator().manual_seed(M + K)
    a = (torch.randn(M, N, generator=g) * 1e-5).cuda()
    b = (torch.randn(M, K, generator=g) * (3.0 if scale == 16.0 else 90.0)).cuda()
    s = (torch.rand(N, generator=g) + 0.5).cuda()
    sc = ops.split_scale_from_amax(a)
    want = ops.gemm_tn_split(a, b, s, sc, scale)
    bs = ops.split_pack(b, scale)
    assert bs.scale == scale and bs.data.shape == b.shape
    ops.split_overflow_reset(a.device)
    got = ops.gemm_tn_split(a, bs.data, s, sc, scale, b_is_split=True)
    assert not ops.split_overflow_raised(a.device)
    assert torch.equal(got, want)
    assert torch.equal(ops.gemm_tn_split(a, bs.data, None, sc, scale, b_is_split=True), ops.gemm_tn_split(a, b, None, sc, scale))
    with pytest.raises(ValueError):
        ops.gemm_tn_split(a, b[:, :4].contiguous(), None, sc, scale, b_is_split=True)          # K % 8


@pytest.mark.parametrize("R,Cin,N", [(70, 64, 64), (300, 128, 96), (800, 512, 512)])
def test_winograd_weight_gradient_from_the_forwards_transformed_input(pkg, R, Cin, N):
    """The 3x3 convolution's weight gradient with the Winograd-domain input the FORWARD wrote (the head of its workspace, split
    layout x 0.25) instead of a second input transform: the same bits."""
    ops = pkg.ops
    gen = torch.Generator().manual_seed(R + 5)
    x = torch.relu(torch.randn(R * 49, Cin, generator=gen)).cuda() * 2.0
    w = (torch.randn(N, Cin, 3, 3, generator=gen) * 0.05).cuda()
    gy = (torch.randn(R * 49, N, generator=gen) * 2e-5).cuda()
    s = (torch.rand(N, generator=gen) + 0.5).cuda()
    U = ops.split_pack(ops.winograd_pack_weight(w))
    ws = torch.empty(ops.winograd_workspace_bytes(R, Cin, N), dtype=torch.uint8, device="cuda")
    y_kept = ops.winograd_conv3x3(x, U, relu=True, roi_major=True, in_roi_major=True, workspace=ws)
    y = ops.winograd_conv3x3(x, U, relu=True, roi_major=True, in_roi_major=True)
    assert torch.equal(y_kept, y)
    want = ops.winograd_wgrad(x, gy, s, roi_major=True, split=True)
    got = ops.winograd_wgrad(x, gy, s, roi_major=True, split=True, v_split=ws)
    assert torch.equal(got, want)
    with pytest.raises(ValueError):
        ops.winograd_wgrad(x, gy, s, roi_major=True, split=False, v_split=ws)
    with pytest.raises(ValueError):
        ops.winograd_conv3x3(x, U, roi_major=True, in_roi_major=True, workspace=ws[:1000])


def test_training_step_with_the_kept_transformed_input_is_the_same_bits(pkg, oracle, monkeypatch):
    """One joint LSM-shaped Res5 step (whole grid + proposals) with the forward's Winograd-domain inputs kept for the weight gradients
    (default) and with the second transform (LOCOV_RES5_KEEP_V=0): every output and gradient equal bit for bit, and the kept path is taken."""
    from locov_amd import res5_train
    R, (in_ch, mid, out_ch) = 21, (128, 64, 256)
    outs = {}
    for keep in (True, False):
        monkeypatch.setattr(res5_train, "_KEEP_V", keep)
        res5, params = _stage(pkg, oracle, in_ch, mid, out_ch, seed=8)
        gen = torch.Generator().manual_seed(29)
        N, H, W = 2, 26, 43
        feat = torch.randn(N, in_ch, H, W, generator=gen).cuda().requires_grad_(True)
        wh = torch.rand(R, 2, generator=gen) * 300 + 30
        xy = torch.rand(R, 2, generator=gen) * 300
        rois = torch.cat([torch.randint(0, N, (R, 1), generator=gen).float(), xy, xy + wh], dim=1).cuda()
        nhwc = res5_train.to_nhwc(feat)
        st = res5_train.Res5Step(res5, True, feat.device, N * 13 * 22 + 49 * R)
        rows = res5_train.grid_segment(st, nhwc)
        x0 = res5_train.roi_segment(st, nhwc, rois, 14, 1.0 / 16, 0, True)
        assert len(st.wino_ws) == (3 if keep else 0)
        grid, box = st.outputs([rows, x0], [False, True])
        (grid.square().mean() + box.sum()).backward()
        torch.cuda.synchronize()
        named = dict(res5.named_parameters())
        outs[keep] = (grid.detach().clone(), box.detach().clone(), {k: named[k].grad.clone() for k in _weight_keys(params)})
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    for k in outs[True][2]:
        assert torch.equal(outs[True][2][k], outs[False][2][k]), k


def test_linear_split_ex_mask_and_device_scale(pkg):
    ops = pkg.ops
    g = torch.Generator().manual_seed(9)
    for (M, N, K) in ((300, 96, 64), (1000, 260, 128), (2049, 512, 2048)):
        x = (torch.randn(M, K, generator=g) * 3e-6).cuda()
        w = (torch.randn(N, K, generator=g) * 0.05).cuda()
        res = (torch.randn(M, N, generator=g) * 1e-6).cuda()
        act = torch.randn(M, N, generator=g).cuda()
        want = x.double() @ w.double().t() + res.double()
        want = torch.where(act.double() > 0, want, torch.zeros_like(want))
        got = ops.linear_split_ex(x, ops.split_pack(w), residual=res, mask=act, x_scale_dev=ops.split_scale_from_amax(x))
        assert rel_err(got, want) < 3e-6
        assert float(got[act <= 0].abs().max()) == 0.0
        # a fixed x_scale of 16 would lose these tiny values to fp16's fixed-point floor: the device scale is what keeps 22 bits
        coarse = ops.linear_split_ex(x, ops.split_pack(w), residual=res, mask=act, x_scale=16.0)
        assert rel_err(coarse, want) > 10 * rel_err(got, want)


@pytest.mark.parametrize("R,Cin,N", [(70, 64, 64), (300, 128, 96)])
def test_winograd_gradients_split_vs_float64(pkg, R, Cin, N):
    ops = pkg.ops
    gen = torch.Generator().manual_seed(R + 1)
    x = torch.relu(torch.randn(R, Cin, 7, 7, generator=gen))
    w = torch.randn(N, Cin, 3, 3, generator=gen) * 0.1
    s = torch.rand(N, generator=gen) + 0.5
    gy = torch.randn(R, N, 7, 7, generator=gen) * 2e-5
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    (F.conv2d(xd, wd, padding=1) * s.double().view(1, -1, 1, 1) * gy.double()).sum().backward()
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(R * 49, -1).contiguous().cuda()
    dw = ops.winograd_wgrad(rows(x), rows(gy), s.cuda(), roi_major=True, split=True)
    assert rel_err(dw, wd.grad) < 3e-5
    if N % 32 == 0:
        act = torch.randn(R, Cin, 7, 7, generator=gen)
        U = ops.split_pack(ops.winograd_pack_weight(ops.conv3x3_weight_flip(w.cuda(), s.cuda())))
        gx = ops.winograd_conv3x3_split_ex(rows(gy), U, mask=rows(act), roi_major=True)
        want = torch.where(act.double() > 0, xd.grad, torch.zeros_like(xd.grad))
        assert rel_err(gx.view(R, 7, 7, Cin).permute(0, 3, 1, 2), want) < 3e-5
