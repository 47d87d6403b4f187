"""Independent pins for the oracle's NMS restatement ([D2-upstream] torchvision.ops.nms / batched_nms as reached from
ovr/modeling/roi_heads/roi_emb_heads.py:280,357).  torchvision is not installable here and the reference holds no
test for it, so the restatement (oracle/lsm_oracle.py: a greedy sweep over index arrays) is checked against a SECOND formulation with a
different structure -- the defining recurrence evaluated on the full IoU matrix with torch ops -- and against the properties greedy NMS
must have whatever the implementation: idempotence, invariance under a permutation of the input, the kept set as an independent
dominating set of the "IoU > threshold" graph, the class-wise form equal to a per-class loop."""
import numpy as np
import pytest
import torch


def _boxes(rng, n, clustered=True):
    c = rng.uniform(0, 400, (max(n // 6, 1), 2)).astype(np.float32)
    ctr = c[rng.integers(0, len(c), n)] + rng.normal(0, 6 if clustered else 80, (n, 2)).astype(np.float32)
    wh = rng.uniform(8, 90, (n, 2)).astype(np.float32)
    return np.concatenate([ctr - wh / 2, ctr + wh / 2], axis=1).astype(np.float32)


def _iou_matrix(b):
    """torch ops, float32, the published formula: inter / (area_i + area_j - inter)."""
    t = torch.from_numpy(b)
    area = (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])
    lt = torch.max(t[:, None, :2], t[None, :, :2])
    rb = torch.min(t[:, None, 2:], t[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return (inter / (area[:, None] + area[None, :] - inter)).numpy()


def _nms_by_definition(boxes, scores, thresh):
    """keep[i] (i in descending-score order, stable) <=> no kept j before i has IoU(j, i) > thresh."""
    order = np.argsort(-scores, kind="stable")
    iou = _iou_matrix(boxes)[np.ix_(order, order)]
    keep = np.zeros(len(order), bool)
    for i in range(len(order)):
        keep[i] = not np.any(keep[:i] & (iou[:i, i] > thresh))
    return order[keep]


@pytest.mark.parametrize("n,thresh,seed", [(1, 0.5, 0), (2, 0.5, 1), (60, 0.5, 2), (300, 0.5, 3), (300, 0.3, 4), (300, 0.7, 5), (257, 0.0, 6)])
def test_nms_equals_its_definition_on_the_iou_matrix(oracle, n, thresh, seed):
    rng = np.random.default_rng(seed)
    boxes = _boxes(rng, n)
    scores = rng.random(n).astype(np.float32)
    if n > 10:
        scores[5] = scores[9]                                # a tie: the stable order decides
        boxes[7] = boxes[3]                                  # duplicates: IoU exactly 1
    got = oracle.nms(boxes, scores, thresh)
    want = _nms_by_definition(boxes, scores, thresh)
    np.testing.assert_array_equal(got, want)
    # properties of ANY greedy NMS
    iou = _iou_matrix(boxes)
    kept = set(got.tolist())
    assert all(not (iou[i, j] > thresh) for i in kept for j in kept if i != j), "kept boxes must not suppress each other"
    rank = {int(i): r for r, i in enumerate(np.argsort(-scores, kind="stable"))}
    for j in range(n):
        if j not in kept:
            assert any(iou[i, j] > thresh and rank[i] < rank[j] for i in kept), "a dropped box is suppressed by a higher-ranked kept box"
    np.testing.assert_array_equal(oracle.nms(boxes[got], scores[got], thresh), np.arange(len(got)))          # idempotent
    if len(np.unique(scores)) == n:                          # distinct scores: the result does not depend on the input order
        perm = rng.permutation(n)
        np.testing.assert_array_equal(np.sort(perm[oracle.nms(boxes[perm], scores[perm], thresh)]), np.sort(got))


def test_nms_threshold_is_a_strict_comparison(oracle):
    """Two unit-height boxes overlapping by exactly half: IoU = 1/3.  Suppressed for thresholds below 1/3, kept AT 1/3 (`>`)."""
    boxes = np.array([[0, 0, 2, 1], [1, 0, 3, 1]], np.float32)
    scores = np.array([0.9, 0.8], np.float32)
    third = np.float32(1.0) / np.float32(3.0)
    assert float(_iou_matrix(boxes)[0, 1]) == float(third)
    np.testing.assert_array_equal(oracle.nms(boxes, scores, float(third)), [0, 1])
    np.testing.assert_array_equal(oracle.nms(boxes, scores, float(np.nextafter(third, np.float32(0)))), [0])


def test_batched_nms_equals_a_per_class_loop(oracle):
    rng = np.random.default_rng(11)
    n = 400
    boxes = _boxes(rng, n)
    scores = rng.random(n).astype(np.float32)
    idxs = rng.integers(0, 7, n)
    got = oracle.batched_nms(boxes, scores, idxs, 0.5)
    want = np.concatenate([np.flatnonzero(idxs == c)[oracle.nms(boxes[idxs == c], scores[idxs == c], 0.5)] for c in range(7)])
    np.testing.assert_array_equal(np.sort(got), np.sort(want))
    assert np.all(np.diff(scores[got]) <= 0), "results come in descending score order"
    # ... and the host-side mirror of the plugin (the device path's reference in tests/test_gpu_kernels.py) agrees with both
    from locov_amd.roi_heads import box_emb_head as beh
    mine = beh.batched_nms(torch.from_numpy(boxes), torch.from_numpy(scores), torch.from_numpy(idxs), 0.5).numpy()
    np.testing.assert_array_equal(mine, got)
