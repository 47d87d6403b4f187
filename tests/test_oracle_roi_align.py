"""Analytic pins for the ROIAlign / level-assignment restatement (SURVEY.md section 7
"hard parts": no torchvision here, so the oracle is checked against closed forms and an
independent pure-numpy evaluation)."""
import math

import numpy as np
import torch


def _ramp(N, C, H, W, ax, ay, c0):
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    f = np.zeros((N, C, H, W), np.float32)
    for n in range(N):
        for c in range(C):
            f[n, c] = ax * xs + ay * ys + c0 + c + 10 * n
    return f


def test_constant_map(oracle):
    f = np.full((1, 3, 20, 30), 2.5, np.float32)
    rois = np.array([[0, 16, 16, 200, 150], [0, 0, 0, 16, 16], [0, 100.3, 20.7, 411.9, 300.1]], np.float32)
    out = oracle.roi_align(f, rois, (7, 7), 1 / 16, 0, True)
    np.testing.assert_allclose(out, 2.5, rtol=1e-6)


def test_linear_ramp_gives_bin_centre_value(oracle):
    # bilinear interpolation reproduces a linear map exactly -> the average over a regular
    # sampling grid equals the value at the bin centre (box fully inside the map).
    H, W = 40, 50
    ax, ay, c0 = 0.25, -0.5, 3.0
    f = _ramp(2, 2, H, W, ax, ay, c0)
    rois = np.array([[0, 40, 48, 360, 400], [1, 100, 80, 164, 112], [1, 33.3, 47.1, 517.9, 333.3]], np.float32)
    P, s = 14, 1 / 16
    out = oracle.roi_align(f, rois, (P, P), s, 0, True)
    for r, roi in enumerate(rois):
        b = int(roi[0])
        x0, y0, x1, y1 = (roi[1:] * s - 0.5).astype(np.float64)
        bw, bh = (x1 - x0) / P, (y1 - y0) / P
        for c in range(2):
            cx = x0 + (np.arange(P) + 0.5) * bw
            cy = y0 + (np.arange(P) + 0.5) * bh
            want = ax * cx[None, :] + ay * cy[:, None] + c0 + c + 10 * b
            np.testing.assert_allclose(out[r, c], want, atol=2e-4)


def test_box_outside_map_is_zero(oracle):
    f = np.ones((1, 2, 10, 10), np.float32)
    rois = np.array([[0, 400, 400, 500, 500], [0, -900, -900, -700, -700]], np.float32)
    out = oracle.roi_align(f, rois, (7, 7), 1 / 16, 0, True)
    assert np.all(out == 0)


def test_aligned_half_pixel_shift(oracle):
    # aligned=True on box b  ==  aligned=False on the same box shifted by -0.5 px in feature space
    # (when the legacy >=1 clamp does not bite, i.e. box larger than 1 feature pixel)
    rng = np.random.default_rng(0)
    f = rng.standard_normal((1, 4, 30, 30)).astype(np.float32)
    roi = np.array([[0, 64, 80, 320, 336]], np.float32)
    a = oracle.roi_align(f, roi, (7, 7), 1 / 16, 2, True)
    shifted = roi.copy()
    shifted[:, 1:] -= 8.0            # 0.5 feature px * stride 16
    b = oracle.roi_align(f, shifted, (7, 7), 1 / 16, 2, False)
    np.testing.assert_allclose(a, b, atol=1e-6)


def test_degenerate_and_ragged(oracle):
    f = np.ones((2, 1, 8, 8), np.float32)
    assert oracle.roi_align(f, np.zeros((0, 5), np.float32), (7, 7), 1 / 16).shape == (0, 1, 7, 7)
    # zero-area and inverted boxes with aligned=True: grid = ceil(<=0) -> no samples -> 0
    rois = np.array([[1, 32, 32, 32, 32], [0, 64, 64, 32, 32]], np.float32)
    out = oracle.roi_align(f, rois, (7, 7), 1 / 16, 0, True)
    assert np.all(out == 0)
    # legacy mode forces them to 1x1 feature px instead
    out = oracle.roi_align(f, rois[:1], (7, 7), 1 / 16, 0, False)
    np.testing.assert_allclose(out, 1.0)


def _numpy_roi_align(f, rois, P, s, sr, aligned):
    """Independent float64 evaluation with np broadcasting (different code path from the C)."""
    N, C, H, W = f.shape
    out = np.zeros((len(rois), C, P, P))
    for r, roi in enumerate(rois):
        b = int(roi[0])
        off = 0.5 if aligned else 0.0
        x0, y0, x1, y1 = [np.float32(np.float32(v) * np.float32(s)) - np.float32(off) for v in roi[1:]]
        rw, rh = np.float32(x1 - x0), np.float32(y1 - y0)
        if not aligned:
            rw, rh = max(rw, np.float32(1)), max(rh, np.float32(1))
        gh = sr if sr > 0 else int(math.ceil(rh / np.float32(P)))
        gw = sr if sr > 0 else int(math.ceil(rw / np.float32(P)))
        bh, bw = np.float32(rh / np.float32(P)), np.float32(rw / np.float32(P))
        for ph in range(P):
            for pw in range(P):
                acc = np.zeros(C)
                for iy in range(gh):
                    y = np.float32(y0 + np.float32(ph * bh)) + np.float32(np.float32((iy + 0.5) * bh) / np.float32(gh))
                    for ix in range(gw):
                        x = np.float32(x0 + np.float32(pw * bw)) + np.float32(np.float32((ix + 0.5) * bw) / np.float32(gw))
                        if y < -1 or y > H or x < -1 or x > W:
                            continue
                        yy, xx = max(float(y), 0.0), max(float(x), 0.0)
                        yl, xl = int(yy), int(xx)
                        if yl >= H - 1:
                            yl = yh = H - 1
                            yy = float(yl)
                        else:
                            yh = yl + 1
                        if xl >= W - 1:
                            xl = xh = W - 1
                            xx = float(xl)
                        else:
                            xh = xl + 1
                        ly, lx = yy - yl, xx - xl
                        acc += ((1 - ly) * (1 - lx) * f[b, :, yl, xl] + (1 - ly) * lx * f[b, :, yl, xh]
                                + ly * (1 - lx) * f[b, :, yh, xl] + ly * lx * f[b, :, yh, xh])
                out[r, :, ph, pw] = acc / max(gh * gw, 1)
    return out


def test_against_independent_numpy_eval(oracle):
    rng = np.random.default_rng(5)
    f = rng.standard_normal((2, 3, 25, 42)).astype(np.float32)
    boxes = oracle.synth_boxes(rng, 24, 42 * 16.0, 25 * 16.0)
    # a few boxes that stick out of the map (unclipped) to exercise the border rules
    boxes[:4] += np.array([-90, -70, 60, 80], np.float32)
    rois = np.concatenate([rng.integers(0, 2, (24, 1)).astype(np.float32), boxes], axis=1)
    for sr, aligned in ((0, True), (2, True), (0, False)):
        got = oracle.roi_align(f, rois, (7, 7), 1 / 16, sr, aligned)
        want = _numpy_roi_align(f, rois, 7, 1 / 16, sr, aligned)
        np.testing.assert_allclose(got, want, atol=2e-5)


def test_backward_is_adjoint_of_forward(oracle):
    rng = np.random.default_rng(6)
    f = rng.standard_normal((2, 2, 12, 16)).astype(np.float32)
    boxes = oracle.synth_boxes(rng, 9, 16 * 16.0, 12 * 16.0)
    rois = np.concatenate([rng.integers(0, 2, (9, 1)).astype(np.float32), boxes], axis=1)
    g = rng.standard_normal((9, 2, 7, 7)).astype(np.float32)
    y = oracle.roi_align(f, rois, (7, 7), 1 / 16, 0, True)
    gf = oracle.roi_align_backward(g, f.shape, rois, 1 / 16, 0, True)
    # <g, A f> == <A^T g, f>
    np.testing.assert_allclose((g.astype(np.float64) * y).sum(), (gf.astype(np.float64) * f).sum(), rtol=1e-4)


def test_level_assignment_matches_torch_formula(oracle):
    """[D2-upstream] assign_boxes_to_levels written with torch CPU ops, as Detectron2 does."""
    rng = np.random.default_rng(7)
    boxes = oracle.synth_boxes(rng, 20000)
    # exact level boundaries: sqrt(area) = 224 * 2^k / 2  etc.
    for k, side in enumerate((56.0, 112.0, 224.0, 448.0, 896.0)):
        boxes[k] = (10, 10, 10 + side, 10 + side)
    boxes[5] = (5, 5, 5, 5)                   # zero area -> log2(1e-8) -> clamps to min level
    got = oracle.assign_boxes_to_levels(boxes, 2, 5, 224, 4)
    t = torch.from_numpy(boxes)
    sizes = torch.sqrt((t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1]))
    lv = torch.floor(4 + torch.log2(sizes / 224 + 1e-8))
    lv = torch.clamp(lv, min=2, max=5).to(torch.int64) - 2
    mism = np.nonzero(got != lv.numpy())[0]
    assert len(mism) == 0, (mism[:10], got[mism[:10]], lv.numpy()[mism[:10]])
    assert got[5] == 0 and got.min() == 0 and got.max() == 3
    assert list(got[:5]) == [0, 1, 2, 3, 3]


def test_roi_pooler_multilevel_scatter(oracle):
    rng = np.random.default_rng(8)
    feats = [rng.standard_normal((2, 3, 64 >> i, 96 >> i)).astype(np.float32) for i in range(4)]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    box_lists = [oracle.synth_boxes(rng, 11, 384.0, 256.0), oracle.synth_boxes(rng, 6, 384.0, 256.0)]
    out = oracle.roi_pooler(feats, box_lists, 7, scales, 0, "ROIAlignV2")
    rois = oracle.boxes_to_pooler_format(box_lists)
    assert rois.shape == (17, 5) and list(rois[:, 0]) == [0.0] * 11 + [1.0] * 6
    lv = oracle.assign_boxes_to_levels(rois[:, 1:], 2, 5)
    for i in range(17):
        want = oracle.roi_align(feats[lv[i]], rois[i:i + 1], (7, 7), scales[lv[i]], 0, True)
        np.testing.assert_array_equal(out[i:i + 1], want)


def test_roi_align_matches_torch_grid_sample_on_interior_boxes(oracle):
    """Third-party pin of the bilinear sampling: torch.nn.functional.grid_sample(mode="bilinear", align_corners=True)
    interpolates at index coordinates with the pixel centres on the integers -- the same samples torchvision's
    roi_align(aligned=True) takes -- so for boxes whose every sample lies inside [0, H-1] x [0, W-1] (no border rules
    involved) the oracle's output must equal the mean of grid_sample's values over each bin's adaptive sampling grid.
    grid_sample is PyTorch's own kernel, not this repository's arithmetic."""
    rng = np.random.default_rng(2024)
    N, C, H, W, P, s = 2, 5, 50, 84, 14, 1.0 / 16
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    rois = []
    while len(rois) < 120:
        b = rng.integers(0, N)
        w = 2.0 ** rng.uniform(4.0, np.log2(700.0))
        h = 2.0 ** rng.uniform(4.0, np.log2(600.0))
        x0 = rng.uniform(8.0, 16 * (W - 1) + 8 - w) if 16 * (W - 1) - w > 0 else None
        y0 = rng.uniform(8.0, 16 * (H - 1) + 8 - h) if 16 * (H - 1) - h > 0 else None
        if x0 is None or y0 is None:
            continue
        rois.append([b, x0, y0, x0 + w, y0 + h])
    rois = np.asarray(rois, np.float32)
    got = oracle.roi_align(feat, rois, (P, P), s, 0, True)
    ft = torch.from_numpy(feat).double()
    worst = 0.0
    grids_seen = set()
    for r, roi in enumerate(rois):
        b = int(roi[0])
        # the oracle's (= torchvision's) fp32 box arithmetic decides the sampling grid; the sample positions are
        # then evaluated in float64
        x0, y0, x1, y1 = [np.float32(v) * np.float32(s) - np.float32(0.5) for v in roi[1:]]
        rw, rh = np.float32(x1 - x0), np.float32(y1 - y0)
        bw, bh = np.float32(rw / np.float32(P)), np.float32(rh / np.float32(P))
        gw, gh = int(np.ceil(bw)), int(np.ceil(bh))
        grids_seen.add((gh, gw))
        # sample coordinate, rounded step by step in fp32 as the kernel forms it: (start + p*bin) + ((i + .5)*bin)/grid
        f32 = np.float32
        coord = lambda start, bin_, g: ((start + np.arange(P, dtype=f32)[:, None] * bin_).astype(f32)
                                        + (((np.arange(g, dtype=f32)[None, :] + f32(0.5)) * bin_).astype(f32) / f32(g)).astype(f32)).astype(f32)
        xs, ys = coord(x0, bw, gw).astype(np.float64), coord(y0, bh, gh).astype(np.float64)                 # [P, g]
        assert xs.min() >= 0 and xs.max() <= W - 1 and ys.min() >= 0 and ys.max() <= H - 1, "box generator: not interior"
        gx = torch.from_numpy(2.0 * xs.reshape(-1) / (W - 1) - 1.0)
        gy = torch.from_numpy(2.0 * ys.reshape(-1) / (H - 1) - 1.0)
        grid = torch.stack(torch.meshgrid(gy, gx, indexing="ij")[::-1], dim=-1)[None]                        # [1, P*gh, P*gw, (x, y)]
        samp = torch.nn.functional.grid_sample(ft[b:b + 1], grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        want = samp.view(C, P, gh, P, gw).mean(dim=(2, 4)).numpy()
        worst = max(worst, float(np.abs(got[r] - want).max()))
    assert len(grids_seen) >= 4, grids_seen                    # adaptive grids 1..4 per axis are all exercised
    assert worst <= 2e-6, worst


def test_roi_align_border_rules_match_torch_grid_sample_border_mode(oracle):
    """Third-party pin of ROIAlign's BORDER rules.  torchvision's bilinear_interpolate treats a sample (y, x) as
        outside [-1, H] x [-1, W]            -> contributes 0 (but still counts in the bin's average)
        otherwise                             -> coordinates clamped to [0, H-1] x [0, W-1], then bilinear
    (`if (y <= 0) y = 0`, `if (y_low >= H-1) { y_high = y_low = H-1; y = y_low }`).  The second line is exactly what
    torch.nn.functional.grid_sample(padding_mode="border", align_corners=True) computes, so: evaluate PyTorch's kernel at
    the oracle's own sample coordinates, zero the samples the first line excludes, average over each bin's adaptive grid,
    and the oracle must agree -- on boxes that cross every edge and corner of the map, lie partly beyond [-1, H], and
    sit entirely inside the clamp band (-1, 0) / (H-1, H)."""
    rng = np.random.default_rng(77)
    N, C, H, W, P, s = 2, 4, 50, 84, 14, 1.0 / 16
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    iw, ih = 16.0 * W, 16.0 * H                       # the map covers [0, 1344) x [0, 800) image pixels
    rois = []
    def add(b, x0, y0, x1, y1):
        rois.append([b, x0, y0, x1, y1])
    for k in range(60):                               # boxes straddling each edge / corner, sizes over all adaptive grids
        b = int(rng.integers(0, N))
        w = 2.0 ** rng.uniform(4.0, np.log2(600.0))
        h = 2.0 ** rng.uniform(4.0, np.log2(500.0))
        side = k % 8
        cx = {0: 0.0, 1: iw, 4: 0.0, 5: iw, 6: 0.0, 7: iw}.get(side, rng.uniform(0, iw)) + rng.uniform(-0.4, 0.4) * w
        cy = {2: 0.0, 3: ih, 4: 0.0, 5: ih, 6: ih, 7: 0.0}.get(side, rng.uniform(0, ih)) + rng.uniform(-0.4, 0.4) * h
        add(b, cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2)
    # thin boxes entirely inside the clamp bands (samples in (-1, 0) and (H-1, H) map coordinates), and just beyond them
    for b in range(N):
        add(b, -6.0, 100.0, -1.0, 300.0)              # x in (-0.875, -0.56): clamped to column 0
        add(b, 200.0, -7.5, 420.0, -0.5)              # y band above the map
        add(b, iw - 7.0, 50.0, iw + 6.0, 500.0)       # around the last column: clamp at W-1, some samples in (W-1, W]
        add(b, 100.0, ih - 5.0, 900.0, ih + 7.0)
        add(b, -40.0, -40.0, -17.0, -17.0)            # everything below -1: exact zeros
        add(b, iw + 9.0, 10.0, iw + 60.0, 300.0)      # everything beyond W
        add(b, -30.0, -30.0, iw + 30.0, ih + 30.0)    # the whole map and more (grid 7 x 4)
    rois = np.asarray(rois, np.float32)
    got = oracle.roi_align(feat, rois, (P, P), s, 0, True)
    ft = torch.from_numpy(feat).double()
    f32 = np.float32
    worst, n_clamped, n_zeroed, grids = 0.0, 0, 0, set()
    for r, roi in enumerate(rois):
        b = int(roi[0])
        x0, y0, x1, y1 = [f32(v) * f32(s) - f32(0.5) for v in roi[1:]]
        rw, rh = f32(x1 - x0), f32(y1 - y0)
        bw, bh = f32(rw / f32(P)), f32(rh / f32(P))
        gw, gh = int(np.ceil(bw)), int(np.ceil(bh))
        grids.add((gh, gw))
        coord = lambda start, bin_, g: ((start + np.arange(P, dtype=f32)[:, None] * bin_).astype(f32)
                                        + (((np.arange(g, dtype=f32)[None, :] + f32(0.5)) * bin_).astype(f32) / f32(g)).astype(f32)).astype(f32)
        xs, ys = coord(x0, bw, gw).reshape(-1), coord(y0, bh, gh).reshape(-1)          # fp32, as the kernel forms them
        ok_x = ~((xs < f32(-1.0)) | (xs > f32(W)))
        ok_y = ~((ys < f32(-1.0)) | (ys > f32(H)))
        n_zeroed += int((~ok_x).sum() + (~ok_y).sum())
        n_clamped += int((ok_x & ((xs < 0) | (xs > W - 1))).sum() + (ok_y & ((ys < 0) | (ys > H - 1))).sum())
        gx = torch.from_numpy(2.0 * xs.astype(np.float64) / (W - 1) - 1.0)
        gy = torch.from_numpy(2.0 * ys.astype(np.float64) / (H - 1) - 1.0)
        grid = torch.stack(torch.meshgrid(gy, gx, indexing="ij")[::-1], dim=-1)[None]
        samp = torch.nn.functional.grid_sample(ft[b:b + 1], grid, mode="bilinear", padding_mode="border", align_corners=True)[0]
        keep = torch.from_numpy(ok_y[:, None] & ok_x[None, :])
        samp = samp * keep.to(samp.dtype)                                              # the [-1, H] rule: excluded samples add 0
        want = samp.view(C, P, gh, P, gw).sum(dim=(2, 4)).numpy() / float(max(gh * gw, 1))
        worst = max(worst, float(np.abs(got[r] - want).max()))
    assert n_clamped > 300 and n_zeroed > 300, (n_clamped, n_zeroed)                   # both rules are exercised, heavily
    assert len(grids) >= 6, grids
    assert worst <= 2e-6, worst
