"""GPU parity of the LSM GroundingHead: the fused alignment kernel and the host
module against (a) vectors recorded from the reference's own GroundingHead.forward
(tests/golden/g4_grounding_head.npz), (b) the oracle at the reference's real sizes, and (c) a
float64 torch re-statement for the gradients."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gh_mod():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device")
    from locov_amd import _lib, grounding_head
    _lib.load()
    return grounding_head


def _cfg(distill=True):
    ns = types.SimpleNamespace
    g = ns(LOCAL_METRIC="dot", GLOBAL_METRIC="aligned_local", ALIGNMENT="softmax", ALIGNMENT_TEMPERATURE=10.0,
           LOSS="cross_entropy", NEGATIVE_MINING="random", TRIPLET_MARGIN=1.0, ALIGN_WORDS_TO_REGIONS=True,
           ALIGN_REGIONS_TO_WORDS=True, TEXT_INPUT="input_embeddings")
    return ns(MODEL=ns(MMSS_HEAD=ns(GROUNDING=g, DISTILLATION_LOSS=distill)))


def _inputs(d, prefix=""):
    img = {"region_features": torch.from_numpy(d[prefix + "region_features"]).cuda(),
           "region_mask": torch.from_numpy(d[prefix + "region_mask"]).cuda()}
    cap = {"input_embeddings": torch.from_numpy(d[prefix + "input_embeddings"]).cuda(),
           "attention_mask": torch.from_numpy(d[prefix + "attention_mask"]).cuda(),
           "special_tokens_mask": torch.from_numpy(d[prefix + "special_tokens_mask"]).cuda()}
    return img, cap


def test_matches_reference_vectors(gh_mod, golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_grounding_head.npz"))
    head = gh_mod.GroundingHead(_cfg(), 256, 96).cuda()
    with torch.no_grad():
        head.v2l_projection.weight.copy_(torch.from_numpy(g["v2l_w"]))
        head.v2l_projection.bias.copy_(torch.from_numpy(g["v2l_b"]))
    for B in (1, 2, 4):
        p = f"b{B}_"
        img, cap = _inputs(g, p)
        with torch.no_grad():
            info, losses, dist = head(img, cap)
        np.testing.assert_allclose(dist["w2r"].cpu().numpy(), g[p + "w2r"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(dist["r2w"].cpu().numpy(), g[p + "r2w"], rtol=2e-5, atol=2e-6)
        assert list(losses.keys()) == [str(n) for n in g[p + "loss_names"]]
        assert list(info.keys()) == [str(n) for n in g[p + "info_names"]]
        np.testing.assert_allclose([float(v) for v in losses.values()], g[p + "losses"], atol=2e-5)
        np.testing.assert_array_equal([float(v) for v in info.values()], g[p + "info"])


def _synth(rng, B, NR, T, V, L):
    d = {"region_features": rng.standard_normal((B, NR, V)).astype(np.float32),
         "region_mask": np.ones((B, NR), np.uint8),
         "input_embeddings": rng.standard_normal((B, T, L)).astype(np.float32),
         "attention_mask": np.ones((B, T), np.int64), "special_tokens_mask": np.zeros((B, T), np.int64)}
    d["special_tokens_mask"][:, 0] = 1
    for b in range(B):
        n = T - 5 * b - 3
        d["attention_mask"][b, n:] = 0
        d["special_tokens_mask"][b, n - 1:] = 1
    if B > 1:
        d["region_mask"][1, NR // 3:] = 0
    return d


def test_reference_sizes_vs_oracle(gh_mod, oracle):
    """configs/coco_lsm.yaml sizes: 4 images/GPU, <= 100 regions, 70 tokens, 2048 -> 768."""
    rng = np.random.default_rng(1992)
    B, NR, T, V, L = 4, 100, 70, 2048, 768
    d = _synth(rng, B, NR, T, V, L)
    d["region_features"] = np.maximum(d["region_features"], 0)
    w = (rng.standard_normal((L, V)) * 0.01).astype(np.float32)
    b = (rng.standard_normal(L) * 0.01).astype(np.float32)
    d["input_embeddings"] *= 0.05
    head = gh_mod.GroundingHead(_cfg(), V, L).cuda()
    with torch.no_grad():
        head.v2l_projection.weight.copy_(torch.from_numpy(w))
        head.v2l_projection.bias.copy_(torch.from_numpy(b))
        info, losses, dist = head(*_inputs(d))
    wl, wi, w2r, r2w = oracle.grounding_head_forward(d["region_features"], d["region_mask"], d["input_embeddings"],
                                                     d["attention_mask"], d["special_tokens_mask"], w, b, 10.0)
    np.testing.assert_allclose(dist["w2r"].cpu().numpy(), w2r, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dist["r2w"].cpu().numpy(), r2w, rtol=1e-4, atol=1e-5)
    for k, v in losses.items():
        assert abs(float(v) - wl[k]) <= 1e-4, k
    for k, v in info.items():
        assert float(v) == wi[k], k


def _torch_reference(region, rmask, cap, cmask, w, b, temp):
    """float64, differentiable re-statement of grounding_head.py:111-277 (dot / softmax / aligned_local / CE)."""
    B, NR, _ = region.shape
    img = region @ w.t() + b
    sim = torch.einsum("ctl,irl->citr", cap, img) / temp
    valid = (cmask[:, None, :, None] * rmask[None, :, None, :]) > 0
    masked = torch.where(valid, sim, sim.min().detach() - 100.0)
    a_w = torch.softmax(masked, 3) * cmask[:, None, :, None]
    a_r = torch.softmax(masked, 2) * rmask[None, :, None, :]
    nw = cmask.sum(1).clamp(min=1)[:, None]
    nr = rmask.sum(1).clamp(min=1)[None, :]
    w2r = (a_w * -sim).sum((2, 3)) / nw
    r2w = (a_r * -sim).sum((2, 3)) / nr
    ok = (cmask.sum(1)[:, None] > 0) | (rmask.sum(1)[None, :] > 0)          # :232-243
    w2r = torch.where(ok, w2r, w2r.max().detach() + 100.0)
    r2w = torch.where(ok, r2w, r2w.max().detach() + 100.0)
    loss = 0
    for cst in (w2r, r2w):
        loss = loss + torch.diag(-torch.log_softmax(-cst, 0)).mean() + torch.diag(-torch.log_softmax(-cst, 1)).mean()
    return loss, w2r, r2w


@pytest.mark.parametrize("distill", [True, False])            # False: the cross-entropy tail in one launch (ops.grounding_ce)
@pytest.mark.parametrize("empty_case", [False, True])
def test_gradients_match_float64_reference(gh_mod, empty_case, distill):
    rng = np.random.default_rng(7)
    B, NR, T, V, L = 3, 19, 11, 64, 32
    d = _synth(rng, B, NR, T, V, L)
    if empty_case:
        d["region_mask"][2, :] = 0              # an image without regions: uniform-attention branch
        d["attention_mask"][1, :] = 0           # a caption without words
    head = gh_mod.GroundingHead(_cfg(distill), V, L).cuda()
    img, cap = _inputs(d)
    img["region_features"].requires_grad_(True)
    cap["input_embeddings"].requires_grad_(True)
    info, losses = head(img, cap)[:2]
    total = sum(losses.values())
    total.backward()
    # float64 reference on the CPU
    region = torch.from_numpy(d["region_features"]).double().requires_grad_(True)
    capt = torch.from_numpy(d["input_embeddings"]).double().requires_grad_(True)
    w = head.v2l_projection.weight.detach().cpu().double().requires_grad_(True)
    bb = head.v2l_projection.bias.detach().cpu().double().requires_grad_(True)
    cmask = torch.from_numpy(d["attention_mask"] * (1 - d["special_tokens_mask"])).double()
    rmask = torch.from_numpy(d["region_mask"]).double()
    ref, w2r, r2w = _torch_reference(region, rmask, capt, cmask, w, bb, 10.0)
    ref.backward()
    assert abs(float(total) - float(ref)) < 1e-4
    np.testing.assert_allclose(img["region_features"].grad.cpu().numpy(), region.grad.numpy(), atol=2e-6, rtol=1e-3)
    np.testing.assert_allclose(cap["input_embeddings"].grad.cpu().numpy(), capt.grad.numpy(), atol=2e-6, rtol=1e-3)
    np.testing.assert_allclose(head.v2l_projection.weight.grad.cpu().numpy(), w.grad.numpy(), atol=2e-6, rtol=1e-3)
    np.testing.assert_allclose(head.v2l_projection.bias.grad.cpu().numpy(), bb.grad.numpy(), atol=2e-6, rtol=1e-3)


def test_weight_tying_with_emb_pred(gh_mod):
    """distill_prop_mmss_gcnn.py:117-125: emb_pred shares v2l_projection's Parameters; both paths must
    read the live tensors."""
    head = gh_mod.GroundingHead(_cfg(distill=False), 64, 32).cuda()
    lin = torch.nn.Linear(64, 32).cuda()
    lin.weight, lin.bias = head.v2l_projection.weight, head.v2l_projection.bias
    rng = np.random.default_rng(3)
    d = _synth(rng, 2, 9, 8, 64, 32)
    out = head(*_inputs(d))
    assert len(out) == 2                                   # no distributions when DISTILLATION_LOSS is off
    l0 = float(sum(out[1].values()))
    with torch.no_grad():
        lin.weight.mul_(2.0)                               # in-place update through the OTHER module
    l1 = float(sum(head(*_inputs(d))[1].values()))
    assert l0 != l1


def _variant_cfg(over, distill):
    import json
    c = _cfg(distill)
    for k, v in (json.loads(over) if isinstance(over, str) else over).items():
        setattr(c.MODEL.MMSS_HEAD.GROUNDING, k, v)
    return c


def test_other_variants_match_reference_vectors(gh_mod, golden_dir):
    """grounding_head.py:161-343 beyond the LSM configuration -- hardmax alignment, reconstruction_mse, triplet loss with
    hardest / easiest negatives, one alignment direction -- against outputs AND v2l_projection gradients recorded from the
    reference's own GroundingHead (tests/golden/make_golden.py g8)."""
    g = np.load(os.path.join(golden_dir, "g8_grounding_variants.npz"))
    for name in (str(n) for n in g["variants"]):
        over = str(g[name + "_cfg"])
        for B in (1, 3):
            p = f"{name}_b{B}_"
            both = (p + "w2r") in g
            head = gh_mod.GroundingHead(_variant_cfg(over, both), 128, 64).cuda()
            with torch.no_grad():
                head.v2l_projection.weight.copy_(torch.from_numpy(g["v2l_w"]))
                head.v2l_projection.bias.copy_(torch.from_numpy(g["v2l_b"]))
            res = head(*_inputs(g, f"b{B}_"))
            info, losses = res[0], res[1]
            assert list(losses.keys()) == [str(n) for n in g[p + "loss_names"]], name
            assert list(info.keys()) == [str(n) for n in g[p + "info_names"]], name
            np.testing.assert_allclose([float(v.detach()) for v in losses.values()], g[p + "losses"], rtol=2e-5, atol=2e-5, err_msg=p)
            np.testing.assert_array_equal([float(v) for v in info.values()], g[p + "info"], err_msg=p)
            if both:
                np.testing.assert_allclose(res[2]["w2r"].detach().cpu().numpy(), g[p + "w2r"], rtol=2e-5, atol=2e-5, err_msg=p)
                np.testing.assert_allclose(res[2]["r2w"].detach().cpu().numpy(), g[p + "r2w"], rtol=2e-5, atol=2e-5, err_msg=p)
            sum(losses.values()).backward()
            want = g[p + "grad_v2l_w"]
            got = head.v2l_projection.weight.grad.cpu().numpy()
            assert np.abs(got - want).max() <= 2e-5 * max(np.abs(want).max(), 1e-3), (p, np.abs(got - want).max(), np.abs(want).max())


def test_random_alignments_and_error_behaviour(gh_mod):
    """random_categorical / random_top3 (:175-206) draw their alignment: with a near-zero temperature the categorical draw IS
    the arg-max, so it must reproduce hardmax; top-3 draws stay inside the three best.  Triplet with random negatives (:297-304)
    lies between the hardest and the easiest choice.  What the reference rejects is rejected the same way."""
    rng = np.random.default_rng(5)
    d = _synth(rng, 3, 9, 8, 64, 32)
    def run(**over):
        torch.manual_seed(0)
        head = gh_mod.GroundingHead(_variant_cfg(over, True), 64, 32).cuda()
        torch.manual_seed(1)
        with torch.no_grad():
            return head(*_inputs(d))
    hard = run(ALIGNMENT="hardmax", ALIGNMENT_TEMPERATURE=1e-6)
    cat = run(ALIGNMENT="random_categorical", ALIGNMENT_TEMPERATURE=1e-6)
    # (a caption without a single valid word has nothing to arg-max over: its alignment is a uniform draw in the reference too)
    has_words = torch.from_numpy((d["attention_mask"] * (1 - d["special_tokens_mask"])).sum(1) > 0).cuda()
    assert bool(has_words.any()) and not bool(has_words.all())
    for k in ("w2r", "r2w"):
        a, b = hard[2][k][has_words], cat[2][k][has_words]
        assert bool(((a - b).abs() <= 1e-5 * a.abs().clamp_min(1.0)).all()), k
    top3 = run(ALIGNMENT="random_top3")
    assert all(torch.isfinite(v) for v in top3[1].values()) and top3[2]["w2r"].shape == (3, 3)
    # aligned-local cost of a top-3 draw can never beat the arg-max alignment's (it is the smallest achievable distance)
    best = run(ALIGNMENT="hardmax")
    # (where there are three valid candidates to draw from: every image here has >= 3 regions, captions need >= 3 words)
    three_words = torch.from_numpy((d["attention_mask"] * (1 - d["special_tokens_mask"])).sum(1) >= 3).cuda()
    assert bool(three_words.any())
    assert bool((top3[2]["w2r"] >= best[2]["w2r"] - 1e-5).all())
    assert bool((top3[2]["r2w"][three_words] >= best[2]["r2w"][three_words] - 1e-5).all())
    t_h, t_e, t_r = (run(LOSS="triplet", NEGATIVE_MINING=m)[1] for m in ("hardest", "easiest", "random"))
    for k in t_r:
        assert float(t_e[k]) - 1e-6 <= float(t_r[k]) <= float(t_h[k]) + 1e-6, k
    with pytest.raises(NotImplementedError):
        gh_mod.GroundingHead(_variant_cfg(dict(LOCAL_METRIC="cosine"), False), 64, 32).cuda()(*_inputs(d))
    with pytest.raises(NotImplementedError):
        gh_mod.GroundingHead(_variant_cfg(dict(ALIGNMENT="optimal_transport"), False), 64, 32).cuda()(*_inputs(d))
    with pytest.raises(NotImplementedError):
        gh_mod.GroundingHead(_variant_cfg(dict(GLOBAL_METRIC="emd"), False), 64, 32).cuda()(*_inputs(d))
    with pytest.raises(Exception, match="Matching loss is not defined"):
        gh_mod.GroundingHead(_variant_cfg(dict(LOSS="matching"), False), 64, 32).cuda()(*_inputs(d))


@pytest.mark.parametrize("words,regions", [(True, True), (True, False), (False, True)])
def test_one_launch_ce_tail_equals_the_torch_ops(gh_mod, golden_dir, words, regions):
    """GroundingHead without the distillation outputs runs grounding_head.py:239-290,357-377 as ONE launch (ops.grounding_ce =
    locov_grounding_ce_fwd / _bwd); with them it keeps the torch ops.  Same names in the same order, same values, same gradients --
    both alignments, either one alone, a caption without words next to an image without regions (the max + 100 pairs), and the
    reference's own vectors."""
    rng = np.random.default_rng(11)
    B, NR, T, V, L = 6, 23, 9, 48, 32
    d = _synth(rng, B, NR, T, V, L)
    d["attention_mask"][:, :4] = 1
    d["special_tokens_mask"][:, 1:4] = 0
    d["region_mask"][4, :] = 0
    d["attention_mask"][3, :] = 0
    outs = []
    for distill in (True, False):
        cfg = _cfg(distill)
        cfg.MODEL.MMSS_HEAD.GROUNDING.ALIGN_WORDS_TO_REGIONS = words
        cfg.MODEL.MMSS_HEAD.GROUNDING.ALIGN_REGIONS_TO_WORDS = regions
        torch.manual_seed(5)
        head = gh_mod.GroundingHead(cfg, V, L).cuda()
        img, cap = _inputs(d)
        img["region_features"].requires_grad_(True)
        cap["input_embeddings"].requires_grad_(True)
        info, losses = head(img, cap)[:2]
        w = torch.linspace(0.5, 2.0, len(losses)).tolist()
        sum(v * k for v, k in zip(losses.values(), w)).backward()
        outs.append((info, losses, img["region_features"].grad, cap["input_embeddings"].grad, head.v2l_projection.weight.grad))
    (ia, la, ga, ca, wa), (ib, lb, gb, cb, wb) = outs
    assert list(la) == list(lb) and list(ia) == list(ib) and len(la) == 2 * (words + regions)
    for k in la:
        assert abs(float(la[k]) - float(lb[k])) <= 1e-6 * max(1.0, abs(float(la[k]))), k
    for k in ia:
        assert float(ia[k]) == float(ib[k]), k
    for x, y in ((ga, gb), (ca, cb), (wa, wb)):
        torch.testing.assert_close(y, x, rtol=1e-4, atol=1e-8)
    if words and regions:
        g = np.load(os.path.join(golden_dir, "g4_grounding_head.npz"))
        head = gh_mod.GroundingHead(_cfg(False), 256, 96).cuda()
        with torch.no_grad():
            head.v2l_projection.weight.copy_(torch.from_numpy(g["v2l_w"]))
            head.v2l_projection.bias.copy_(torch.from_numpy(g["v2l_b"]))
            for Bg in (1, 2, 4):
                p = f"b{Bg}_"
                info, losses = head(*_inputs(g, p))
                assert list(losses.keys()) == [str(n) for n in g[p + "loss_names"]]
                assert list(info.keys()) == [str(n) for n in g[p + "info_names"]]
                np.testing.assert_allclose([float(v) for v in losses.values()], g[p + "losses"], atol=2e-5)
                np.testing.assert_array_equal([float(v) for v in info.values()], g[p + "info"])
