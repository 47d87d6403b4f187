"""f-2 on the device: per-dataset bank swaps during evaluation (ovr/engine/trainer.py:187-191,254-257:
48 / 17 / 65-class COCO banks, and an LVIS-size 1203-class one) through TextBankCache with its pre-packed bf16 copies,
logits against the oracle after every swap; plus the cache-coherence rules of the packed bank and the cls_score bias."""
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


def _predictor(pkg, sim_dtype, c5=256, dim=768):
    cfg = pkg.config.get_cfg()
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    cfg.MODEL.ROI_BOX_HEAD.EMB_DIM = dim
    cfg.MODEL.ROI_BOX_HEAD.SIM_GEMM_DTYPE = sim_dtype
    torch.manual_seed(4)
    return pkg.build_box_predictor(cfg, c5).cuda().eval()


def _bank(rng, k, dim=768):
    m = np.zeros((k + 1, dim), np.float32)
    m[:k] = rng.standard_normal((k, dim)) * 0.05
    return m


@pytest.mark.parametrize("sim_dtype", ["fp32", "bf16"])
def test_bank_swaps_on_the_device_vs_oracle(pkg, oracle, sim_dtype):
    from locov_amd.text_bank import TextBankCache
    rng = np.random.default_rng(12)
    bp = _predictor(pkg, sim_dtype)
    cache = TextBankCache(device="cuda", with_bf16=True)
    banks = {"coco_seen": _bank(rng, 48), "coco_unseen": _bank(rng, 17), "coco_all": _bank(rng, 65), "lvis_v1": _bank(rng, 1203)}
    for name, m in banks.items():
        cache.add(name, m)

    class Heads:
        num_classes = None
    heads = Heads()
    x = rng.standard_normal((300, 256)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    w = {k: v.detach().cpu().numpy() for k, v in bp.state_dict().items()}
    for name in ("coco_unseen", "lvis_v1", "coco_seen", "coco_all", "lvis_v1", "coco_unseen"):
        cache.install(name, bp, heads)
        k = banks[name].shape[0] - 1
        assert bp.num_classes == k == heads.num_classes
        if sim_dtype == "bf16":
            assert bp._packed_bank().data_ptr() == cache.get(name, bf16=True).data_ptr()        # the pre-packed copy, no conversion
        with torch.no_grad():
            scores, deltas = bp(xd)
        want, want_d, emb = oracle.box_predictor_forward(x, w["emb_pred.weight"], w["emb_pred.bias"], w["bbox_pred.weight"],
                                                         w["bbox_pred.bias"], banks[name])
        assert tuple(scores.shape) == (300, k + 1) and torch.all(scores[:, -1] == 0)
        if sim_dtype == "fp32":
            assert np.abs(scores.cpu().numpy() - want).max() <= 1e-4
        else:
            emb_dev = bp.region_embedding(xd)              # emb_pred in the predictor's own inference arithmetic
            want16 = (emb_dev.cpu().to(torch.bfloat16).double() @ torch.from_numpy(banks[name]).to(torch.bfloat16).double().t()).numpy()
            assert np.abs(scores.cpu().numpy() - want16).max() <= 1e-4
            assert np.abs(scores.cpu().numpy() - want).max() <= 5e-2
        np.testing.assert_allclose(deltas.cpu().numpy(), want_d, atol=1e-6)


def test_packed_bank_follows_the_weight_and_bias_is_honoured(pkg, oracle):
    rng = np.random.default_rng(13)
    bp = _predictor(pkg, "bf16")
    bp.set_class_embeddings(_bank(rng, 30))
    x = torch.from_numpy(rng.standard_normal((64, 256)).astype(np.float32)).cuda()
    with torch.no_grad():
        s0, _ = bp(x)
        # same-shape re-assignment / in-place edit of the bank must not leave a stale bf16 copy behind
        new = torch.from_numpy(_bank(rng, 30)).cuda()
        bp.cls_score.weight.data = new
        s1, _ = bp(x)
        bp.sim_gemm_dtype = "fp32"
        s1_f32, _ = bp(x)
        bp.sim_gemm_dtype = "bf16"
        assert (s1 - s1_f32).abs().max() <= 5e-2 and (s1 - s0).abs().max() > 0.05
        bp.cls_score.weight.mul_(2.0)
        s2, _ = bp(x)
        assert (s2 - 2 * s1).abs().max() <= 1e-3 * s1.abs().max()
        # a non-zero cls_score.bias (loaded or edited) is added like the reference's nn.Linear does, on every path
        bp.cls_score.bias.data = torch.linspace(-1, 1, 31, device="cuda")
        s3, _ = bp(x)
        assert (s3 - (s2 + bp.cls_score.bias)).abs().max() <= 1e-6
        bp.sim_gemm_dtype = "fp32"
        s4, _ = bp(x)
        ref = pkg.ops.linear(pkg.ops.linear(x, bp.emb_pred.weight, bp.emb_pred.bias), bp.cls_score.weight, bp.cls_score.bias)
        assert (s4 - ref).abs().max() <= 1e-6


def test_same_size_bank_swaps_never_score_against_the_previous_bank(pkg, oracle):
    """every swap builds a fresh nn.Linear (version 0) and the freed bank's address can be handed to the next
    bank of the same shape -- a (data_ptr, _version) cache key then matches a different tensor.  Two different banks with the
    same K swapped back and forth, on the split-operand similarity GEMM (the cached packing) and the bf16 one."""
    rng = np.random.default_rng(14)
    for sim_dtype in ("fp32", "bf16"):
        bp = _predictor(pkg, sim_dtype)
        assert bp.fc_dtype == "f16x2"
        banks = [_bank(rng, 47), _bank(rng, 47)]
        x = rng.standard_normal((96, 256)).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        w = {k: v.detach().cpu().numpy() for k, v in bp.state_dict().items()}
        seen_ptrs = set()
        for i in range(6):
            m = banks[i % 2]
            bp.set_class_embeddings(m)                       # the previous cls_score (and its weight) is dropped here
            torch.cuda.synchronize()
            seen_ptrs.add(bp.cls_score.weight.data_ptr())
            with torch.no_grad():
                scores, _ = bp(xd)
            want, _, _ = oracle.box_predictor_forward(x, w["emb_pred.weight"], w["emb_pred.bias"], w["bbox_pred.weight"],
                                                      w["bbox_pred.bias"], m)
            tol = 1e-4 if sim_dtype == "fp32" else 5e-2
            assert np.abs(scores.cpu().numpy() - want).max() <= tol, (sim_dtype, i)
        # (the allocator did recycle addresses in this loop on the test box: the stale-key failure was reachable)
        assert len(seen_ptrs) <= 6


def test_nms_per_class_fallback_on_the_device(pkg, monkeypatch):
    from locov_amd.roi_heads import box_emb_head as beh
    g = torch.Generator().manual_seed(21)
    n = 5000
    xy = torch.rand(n, 2, generator=g) * 1000
    wh = torch.rand(n, 2, generator=g) * 120 + 4
    boxes = torch.cat([xy, xy + wh], dim=1).cuda()
    scores = torch.rand(n, generator=g).cuda()
    idxs = torch.randint(0, 40, (n,), generator=g).cuda()
    one = beh.batched_nms(boxes, scores, idxs, 0.5)
    monkeypatch.setattr(beh, "_PER_CLASS_NMS_ABOVE", 1000)
    assert torch.equal(one, beh.batched_nms(boxes, scores, idxs, 0.5))
