#!/usr/bin/env python
"""
Generates tests/golden/*.npz by running the REFERENCE's own code in the build
container (where /root/reference is mounted).  The reference's Python never
travels to the GPU box; only these fixtures (inputs + expected outputs) do.

    python tests/golden/make_golden.py            # needs /root/reference

What runs from the reference (loaded by file path, bypassing ovr/__init__.py,
which imports Detectron2):
  G1  ovr/modeling/logged_module.py  normalize_vec, standardize_vec   (:55-72)
      ovr/misc.py                    l2_normalize                     (:46-59)
  G2  ovr/misc.py                    dot_similarity                   (:5-28)
  G3  ovr/modeling/roi_heads/box_emb_head.py
        EmbeddingFastRCNNOutputLayers.{forward, forward_cls_prediction,
        set_class_embeddings}                                         (:179-236)
  G4  ovr/modeling/mmss_heads/grounding_head.py  GroundingHead.forward (:92-388)
  G8  the same, in the configurations other than the LSM one (hardmax, reconstruction_mse, triplet, one direction)  [make_golden.py g8]
  G5  ovr/modeling/meta_arch/distill_mmss_gcnn.py  MultiDistillLoss, MultiDistillLossJS,
      MultiDistillLossL2 .forward                                     (:211-433)
  G6  ovr/modeling/roi_heads/box_emb_grounding_head.py  GroundingModule.set_class_embeddings /
      .forward (multi-token class scoring)                            (:60-256)
  G7  text bank: the statements of tools/coco_bert_embeddings.py:26-34 (token pooling),
      ovr/data/datasets/coco_instances.py:237-254 and lvis_instances.py:269-278 (class_emb_mtx layout).
      Those statements sit inside a script / inside registration functions that need Detectron2, a BERT
      checkpoint and the COCO / LVIS annotation files, so they are executed IN PLACE: the line ranges are read
      from the reference files at generation time and exec'd on seeded stand-in inputs (the fixture holds only
      inputs and outputs).          python tests/golden/make_golden.py g7   regenerates G7 alone.

Detectron2 / fvcore are not installed, so import-time names are satisfied with
inert stand-ins (below).  The ONLY stand-in whose behaviour reaches a golden
value is `FastRCNNOutputLayers.__init__`, which creates `bbox_pred =
nn.Linear(C, 4)` exactly as Detectron2's class-agnostic predictor does; its
weights are then overwritten by the seeded inputs stored in the fixture.
Everything else (`configurable`, `Registry`, event storage, structures) is never
executed on the recorded paths.  `Tensor.to("cuda")` / `.cuda()` are mapped to
CPU because grounding_head.py hard-codes the device.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REF = os.environ.get("LOCOV_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 1992  # configs/coco_lsm.yaml:126


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_standins():
    class _Storage:
        def put_scalar(self, *a, **k):
            pass

    class Registry:
        def __init__(self, name):
            self._name, self._map = name, {}

        def register(self, obj=None):
            if obj is None:
                def deco(o):
                    self._map[o.__name__] = o
                    return o
                return deco
            self._map[obj.__name__] = obj
            return obj

        def get(self, name):
            return self._map[name]

    def configurable(fn=None, **kw):
        return fn

    class FastRCNNOutputLayers(nn.Module):
        """Stand-in for the Detectron2 base class: class-agnostic bbox_pred only."""

        def __init__(self, input_shape, *, box2box_transform=None, num_classes=0,
                     cls_agnostic_bbox_reg=True, loss_weight=1.0, **kw):
            super().__init__()
            c = input_shape if isinstance(input_shape, int) else input_shape.channels
            self.bbox_pred = nn.Linear(c, 4)
            self.loss_weight = {"loss_cls": 1.0, "loss_box_reg": 1.0}

    class ShapeSpec:
        def __init__(self, channels=None, height=None, width=None, stride=None):
            self.channels, self.height, self.width, self.stride = channels, height, width, stride

    _mod("detectron2")
    _mod("detectron2.utils")
    _mod("detectron2.utils.events", get_event_storage=lambda: _Storage())
    _mod("detectron2.utils.registry", Registry=Registry)
    _mod("detectron2.config", configurable=configurable)
    _mod("detectron2.layers", ShapeSpec=ShapeSpec, batched_nms=None, cat=torch.cat,
         cross_entropy=None, nonzero_tuple=None)
    _mod("detectron2.modeling")
    _mod("detectron2.modeling.box_regression", Box2BoxTransform=object)
    _mod("detectron2.modeling.roi_heads")
    _mod("detectron2.modeling.roi_heads.fast_rcnn", fast_rcnn_inference=None,
         fast_rcnn_inference_single_image=None, FastRCNNOutputLayers=FastRCNNOutputLayers,
         _log_classification_stats=None)
    _mod("detectron2.structures", Boxes=object, Instances=object)
    _mod("fvcore")
    _mod("fvcore.nn", giou_loss=None, smooth_l1_loss=None)
    # package skeleton so that "from ovr.modeling... import" resolves by path
    _mod("ovr")
    _mod("ovr.modeling")
    _mod("ovr.modeling.roi_heads")
    _mod("ovr.modeling.mmss_heads")


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def cuda_to_cpu_shim():
    orig_to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
        if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
            k["device"] = "cpu"
        return orig_to(self, *a, **k)

    torch.Tensor.to = to
    torch.Tensor.cuda = lambda self, *a, **k: self


def _ref_lines(rel, first, last):
    """Source lines first..last (1-based, inclusive) of a reference file, dedented, as a code object."""
    import textwrap
    with open(os.path.join(REF, rel)) as f:
        lines = f.read().splitlines()[first - 1:last]
    return compile(textwrap.dedent("\n".join(lines)), f"{rel}:{first}-{last}", "exec")


def make_g7():
    import json
    rng = np.random.default_rng(SEED)
    g7 = {}
    # --- token pooling: tools/coco_bert_embeddings.py:26-34 on a stand-in for the BERT encoder's output
    K, T, D = 9, 6, 32
    special = np.zeros((K, T), np.int64)
    special[:, 0] = 1
    for k in range(K):
        special[k, 1 + 1 + (k % 4):] = 1               # 1..4 real tokens, then [SEP] / padding
    emb = rng.standard_normal((K, T, D)).astype(np.float32)
    ns = {"torch": torch, "class_list": [f"class {k}" for k in range(K)],
          "encoded_class_list": {"special_tokens_mask": torch.from_numpy(special), "input_embeddings": torch.from_numpy(emb)}}
    exec(_ref_lines("tools/coco_bert_embeddings.py", 26, 34), ns)
    g7["pool_input_embeddings"], g7["pool_special_tokens_mask"] = emb, special
    g7["pool_embeddings"] = ns["embeddings"].numpy()
    g7["pool_json"] = np.asarray(json.dumps(ns["class_name_to_bertemb"]))       # what the script writes to disk
    # --- class_emb_mtx layout, COCO: ovr/data/datasets/coco_instances.py:237-254 (one class with a multi-token [T, D] entry)
    names = [f"noun{k}" for k in range(7)]
    table = {n: rng.standard_normal(D).astype(np.float32).tolist() for n in names}
    table["noun3"] = rng.standard_normal((3, D)).astype(np.float32).tolist()   # multi-token entry -> its row stays zero
    things = ["noun5", "noun0", "noun3", "noun6"]

    class Meta:
        def __init__(self):
            self.thing_classes, self.fields = things, {}

        def set(self, **kw):
            self.fields.update(kw)

    meta = Meta()
    ns = {"np": np, "noun_embeddings": json.loads(json.dumps(table)), "dataset_metadata": meta}
    exec(_ref_lines("ovr/data/datasets/coco_instances.py", 237, 254), ns)
    g7["bank_json"] = np.asarray(json.dumps(table))
    g7["bank_thing_classes"] = np.asarray(things)
    g7["coco_class_emb_mtx"] = meta.fields["class_emb_mtx"]
    g7["coco_multi_token_idx"] = np.asarray(sorted(k for k, v in meta.fields["class_embeddings"].items() if v.ndim == 2))
    g7["coco_multi_token_emb"] = meta.fields["class_embeddings"][2]
    # --- LVIS: ovr/data/datasets/lvis_instances.py:269-278 (1-D entries only)
    things_l = ["noun6", "noun1", "noun2"]
    meta = Meta()
    meta.thing_classes = things_l
    ns = {"np": np, "noun_embeddings": json.loads(json.dumps(table)), "metadata": meta}
    exec(_ref_lines("ovr/data/datasets/lvis_instances.py", 269, 278), ns)
    g7["lvis_thing_classes"] = np.asarray(things_l)
    g7["lvis_class_emb_mtx"] = meta.fields["class_emb_mtx"]
    np.savez_compressed(os.path.join(OUT, "g7_text_bank.npz"), **g7)
    print("wrote g7_text_bank.npz")


G8_VARIANTS = {        # name -> overrides of the LSM grounding configuration (grounding_head.py:161-343)
    "hardmax": dict(ALIGNMENT="hardmax"),
    "recmse": dict(GLOBAL_METRIC="reconstruction_mse", ALIGN_REGIONS_TO_WORDS=False),
    "hardmax_recmse": dict(ALIGNMENT="hardmax", GLOBAL_METRIC="reconstruction_mse", ALIGN_REGIONS_TO_WORDS=False),
    "triplet_hardest": dict(LOSS="triplet", NEGATIVE_MINING="hardest", TRIPLET_MARGIN=0.5),
    "triplet_easiest": dict(LOSS="triplet", NEGATIVE_MINING="easiest", TRIPLET_MARGIN=2.0),
    "words_only": dict(ALIGN_REGIONS_TO_WORDS=False),
    "regions_only_hardmax": dict(ALIGNMENT="hardmax", ALIGN_WORDS_TO_REGIONS=False),
}


def make_g8():
    """G8: GroundingHead.forward (grounding_head.py:92-388) in the configurations OTHER than the LSM one, run from the
    reference on CPU: losses, batch accuracies, the [B,B] cost matrices and the gradient of the summed losses with respect
    to v2l_projection.weight.  (reconstruction_mse with ALIGN_REGIONS_TO_WORDS only broadcasts for B*B == NR in the
    reference, :221-224, so those variants align words only.)"""
    install_standins()
    cuda_to_cpu_shim()
    load("ovr.misc", "ovr/misc.py")
    load("ovr.modeling.logged_module", "ovr/modeling/logged_module.py")
    gh = load("ovr.modeling.mmss_heads.grounding_head", "ovr/modeling/mmss_heads/grounding_head.py")

    class _N(dict):
        __getattr__ = dict.__getitem__

    g = torch.Generator().manual_seed(SEED + 8)
    V, L, T = 128, 64, 10
    w = torch.randn(L, V, generator=g) * 0.05
    b = torch.randn(L, generator=g) * 0.05
    out = dict(seed=SEED + 8, v2l_w=w.numpy(), v2l_b=b.numpy(), variants=np.array(list(G8_VARIANTS)))
    inputs = {}
    for B, NR in ((1, 5), (3, 9)):
        region = torch.randn(B, NR, V, generator=g)
        rmask = torch.ones(B, NR, dtype=torch.uint8)
        if B > 1:
            rmask[1, NR // 2:] = 0
        cap = torch.randn(B, T, L, generator=g)
        att = torch.ones(B, T, dtype=torch.int64)
        spec = torch.zeros(B, T, dtype=torch.int64)
        spec[:, 0] = 1
        for i in range(B):
            n = T - 2 * i
            att[i, n:] = 0
            spec[i, n - 1:] = 1
        inputs[B] = (region, rmask, cap, att, spec)
        p = f"b{B}_"
        out.update({p + "region_features": region.numpy(), p + "region_mask": rmask.numpy(), p + "input_embeddings": cap.numpy(),
                    p + "attention_mask": att.numpy(), p + "special_tokens_mask": spec.numpy()})
    for name, over in G8_VARIANTS.items():
        gcfg = dict(LOCAL_METRIC="dot", GLOBAL_METRIC="aligned_local", ALIGNMENT="softmax", ALIGNMENT_TEMPERATURE=10.0,
                    LOSS="cross_entropy", NEGATIVE_MINING="random", TRIPLET_MARGIN=1.0, ALIGN_WORDS_TO_REGIONS=True,
                    ALIGN_REGIONS_TO_WORDS=True, TEXT_INPUT="input_embeddings")
        gcfg.update(over)
        both = gcfg["ALIGN_WORDS_TO_REGIONS"] and gcfg["ALIGN_REGIONS_TO_WORDS"]
        cfg = _N(MODEL=_N(MMSS_HEAD=_N(GROUNDING=_N(gcfg), DISTILLATION_LOSS=both)))
        out[name + "_cfg"] = np.asarray(json.dumps(gcfg))
        for B, (region, rmask, cap, att, spec) in inputs.items():
            head = gh.GroundingHead(cfg, V, L)
            with torch.no_grad():
                head.v2l_projection.weight.copy_(w)
                head.v2l_projection.bias.copy_(b)
            res = head({"region_features": region, "region_mask": rmask},
                       {"input_embeddings": cap, "attention_mask": att, "special_tokens_mask": spec})
            info, losses = res[0], res[1]
            sum(losses.values()).backward()
            p = f"{name}_b{B}_"
            out[p + "loss_names"] = np.array(list(losses.keys()))
            out[p + "losses"] = np.array([float(v) for v in losses.values()], np.float32)
            out[p + "info_names"] = np.array(list(info.keys()))
            out[p + "info"] = np.array([float(v) for v in info.values()], np.float32)
            out[p + "grad_v2l_w"] = head.v2l_projection.weight.grad.numpy().copy()
            if both:
                out[p + "w2r"] = res[2]["w2r"].detach().numpy()
                out[p + "r2w"] = res[2]["r2w"].detach().numpy()
    np.savez_compressed(os.path.join(OUT, "g8_grounding_variants.npz"), **out)
    print("wrote g8_grounding_variants.npz")


def main():
    assert os.path.isdir(REF), f"reference not found at {REF}"
    if sys.argv[1:] == ["g7"]:
        make_g7()
        return
    if sys.argv[1:] == ["g8"]:
        make_g8()
        return
    install_standins()
    cuda_to_cpu_shim()
    misc = load("ovr.misc", "ovr/misc.py")
    lm = load("ovr.modeling.logged_module", "ovr/modeling/logged_module.py")
    # box_emb_head imports its sibling grounding predictor at module scope
    load("ovr.modeling.roi_heads.box_emb_grounding_head",
         "ovr/modeling/roi_heads/box_emb_grounding_head.py")
    beh = load("ovr.modeling.roi_heads.box_emb_head", "ovr/modeling/roi_heads/box_emb_head.py")
    gh = load("ovr.modeling.mmss_heads.grounding_head", "ovr/modeling/mmss_heads/grounding_head.py")

    g = torch.Generator().manual_seed(SEED)

    # ---- G1: row normalisation ------------------------------------------------
    x = torch.randn(32, 768, generator=g)
    x[7] = 0.0                         # all-zero row
    x[11] *= 1e-20                     # tiny-norm row (exercises the eps clamp)
    x[13] = 3.25                       # constant row (std = 0 under standardise)
    np.savez_compressed(
        os.path.join(OUT, "g1_rownorm.npz"), seed=SEED, x=x.numpy(),
        normalize_vec=lm.normalize_vec(x, dim=1).numpy(),
        standardize_vec=lm.standardize_vec(x, dim=1).numpy(),
        l2_normalize=misc.l2_normalize(x, -1).numpy())

    # ---- G2: dot similarity ---------------------------------------------------
    emb = torch.randn(64, 768, generator=g) * 0.5
    bank81 = torch.randn(81, 768, generator=g) * 0.05
    bank81[-1] = 0
    emb96 = torch.randn(16, 96, generator=g) * 0.5          # LVIS-size bank, small D
    bank1204 = torch.randn(1204, 96, generator=g) * 0.05
    bank1204[-1] = 0
    np.savez_compressed(
        os.path.join(OUT, "g2_dot_similarity.npz"), seed=SEED, emb=emb.numpy(),
        bank81=bank81.numpy(), emb96=emb96.numpy(), bank1204=bank1204.numpy(),
        sim81=misc.dot_similarity(emb, bank81).numpy(),
        sim1204=misc.dot_similarity(emb96, bank1204).numpy())

    # ---- G3: EmbeddingFastRCNNOutputLayers ------------------------------------
    R, C5, D, K = 48, 512, 192, 80   # reduced C5/D keep the fixture small
    feats = torch.randn(R, C5, generator=g).abs()          # post-ReLU means are >= 0
    emb_w = torch.randn(D, C5, generator=g) * 0.01
    emb_b = torch.randn(D, generator=g) * 0.01
    bbox_w = torch.randn(4, C5, generator=g) * 0.001
    bbox_b = torch.randn(4, generator=g) * 0.001
    bank = torch.randn(K + 1, D, generator=g) * 0.05
    bank[-1] = 0
    g3 = dict(seed=SEED, feats=feats.numpy(), emb_w=emb_w.numpy(), emb_b=emb_b.numpy(),
              bbox_w=bbox_w.numpy(), bbox_b=bbox_b.numpy(), bank=bank.numpy())
    for tag, norm, std in (("dot", False, False), ("norm", True, False), ("std", False, True)):
        m = beh.EmbeddingFastRCNNOutputLayers(
            C5, box2box_transform=None, num_classes=K, cls_agnostic_bbox_reg=True,
            emb_dim=D, embedding_based=True, freeze_emb_pred=True, normalize_emb=norm,
            standardize_emb=std, detach_cls_predictor=True)
        with torch.no_grad():
            m.emb_pred.weight.copy_(emb_w)
            m.emb_pred.bias.copy_(emb_b)
            m.bbox_pred.weight.copy_(bbox_w)
            m.bbox_pred.bias.copy_(bbox_b)
        m.set_class_embeddings(bank.numpy())
        assert m.num_classes == K
        scores, deltas = m(feats)
        scores4d, _ = m(feats.reshape(R, C5, 1, 1))          # x.dim() > 2 branch (:189-190)
        assert torch.equal(scores, scores4d)
        g3[f"scores_{tag}"] = scores.detach().numpy()
        g3[f"deltas_{tag}"] = deltas.detach().numpy()
        g3[f"cls_w_{tag}"] = m.cls_score.weight.detach().numpy()
        g3[f"cls_b_{tag}"] = m.cls_score.bias.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "g3_box_predictor.npz"), **g3)

    # ---- G4: GroundingHead.forward --------------------------------------------
    class _N(dict):
        __getattr__ = dict.__getitem__

    cfg = _N(MODEL=_N(MMSS_HEAD=_N(
        GROUNDING=_N(LOCAL_METRIC="dot", GLOBAL_METRIC="aligned_local", ALIGNMENT="softmax",
                     ALIGNMENT_TEMPERATURE=10.0, LOSS="cross_entropy", NEGATIVE_MINING="random",
                     TRIPLET_MARGIN=1.0, ALIGN_WORDS_TO_REGIONS=True, ALIGN_REGIONS_TO_WORDS=True,
                     TEXT_INPUT="input_embeddings"),
        DISTILLATION_LOSS=True)))
    g4 = dict(seed=SEED)
    V, L, T = 256, 96, 12           # small v_dim / l_dim / tokens keep the fixture tiny
    head = gh.GroundingHead(cfg, V, L)
    with torch.no_grad():
        head.v2l_projection.weight.copy_(torch.randn(L, V, generator=g) * 0.05)
        head.v2l_projection.bias.copy_(torch.randn(L, generator=g) * 0.05)
    g4["v2l_w"] = head.v2l_projection.weight.detach().numpy()
    g4["v2l_b"] = head.v2l_projection.bias.detach().numpy()
    for B, NR in ((1, 5), (2, 9), (4, 17)):
        region = torch.randn(B, NR, V, generator=g)
        rmask = torch.ones(B, NR, dtype=torch.uint8)
        if B > 1:
            rmask[1, NR // 2:] = 0                   # ragged regions
        cap = torch.randn(B, T, L, generator=g)
        att = torch.ones(B, T, dtype=torch.int64)
        spec = torch.zeros(B, T, dtype=torch.int64)
        spec[:, 0] = 1                               # [CLS]
        for b in range(B):
            n = T - 2 * b                            # ragged captions
            att[b, n:] = 0
            spec[b, n - 1:] = 1                      # [SEP] + padding are special
        with torch.no_grad():
            info, losses, dist = head(
                {"region_features": region, "region_mask": rmask},
                {"input_embeddings": cap, "attention_mask": att, "special_tokens_mask": spec})
        p = f"b{B}_"
        g4[p + "region_features"] = region.numpy()
        g4[p + "region_mask"] = rmask.numpy()
        g4[p + "input_embeddings"] = cap.numpy()
        g4[p + "attention_mask"] = att.numpy()
        g4[p + "special_tokens_mask"] = spec.numpy()
        g4[p + "w2r"] = dist["w2r"].numpy()
        g4[p + "r2w"] = dist["r2w"].numpy()
        g4[p + "loss_names"] = np.array(list(losses.keys()))
        g4[p + "losses"] = np.array([float(v) for v in losses.values()], np.float32)
        g4[p + "info_names"] = np.array(list(info.keys()))
        g4[p + "info"] = np.array([float(v) for v in info.values()], np.float32)
    np.savez_compressed(os.path.join(OUT, "g4_grounding_head.npz"), **g4)

    # ---- G5: distillation losses over [B,B] cost matrices ---------------------------
    # (module-scope imports of the meta-arch file are satisfied with inert names; only the three
    # loss classes, plain torch arithmetic, are executed)
    _mod("detectron2.modeling.backbone", build_backbone=None)
    sys.modules["detectron2.modeling"].META_ARCH_REGISTRY = sys.modules["detectron2.utils.registry"].Registry("META_ARCH")
    sys.modules["detectron2.structures"].ImageList = object
    _mod("ovr.modeling.language")
    _mod("ovr.modeling.language.backbone", build_backbone=None)
    _mod("ovr.modeling.mmss_heads.mmss_heads", build_mmss_heads=None)
    _mod("ovr.modeling.meta_arch")
    dm = load("ovr.modeling.meta_arch.distill_mmss_gcnn", "ovr/modeling/meta_arch/distill_mmss_gcnn.py")
    g5 = {}
    case = 0
    for B in (2, 4, 7):
        for temp in (1.0, 2.5):
            trans = torch.randn(B, B, generator=g) * 3.0
            w2r = torch.randn(B, B, generator=g) * 2.0 + 1.0
            r2w = torch.randn(B, B, generator=g) * 0.5 - 1.0
            for name in ("MultiDistillLoss", "MultiDistillLossJS", "MultiDistillLossL2"):
                for tt in (True, False):
                    mod = getattr(dm, name)(temp, loss_weight=0.7, detach_teacher=True, transformer_teacher=tt)
                    g5[f"c{case}_{name}_tt{int(tt)}"] = np.float32(mod(trans, w2r, r2w))
            g5[f"c{case}_trans"], g5[f"c{case}_w2r"], g5[f"c{case}_r2w"] = trans.numpy(), w2r.numpy(), r2w.numpy()
            g5[f"c{case}_temp"] = np.float32(temp)
            case += 1
    g5["num_cases"] = np.int64(case)
    np.savez_compressed(os.path.join(OUT, "g5_distill_losses.npz"), **g5)

    # ---- G6: multi-token grounding predictor ---------------------------------------
    begh = sys.modules["ovr.modeling.roi_heads.box_emb_grounding_head"]
    D6, K6, R6 = 48, 9, 37
    ntok = [1, 3, 2, 5, 1, 4, 2, 1, 3]
    embs = {k: torch.randn(n, D6, generator=g) * 0.3 for k, n in enumerate(ntok)}
    img = torch.randn(R6, D6, generator=g)
    g6 = {"ntok": np.array(ntok, np.int32), "image_emb": img.numpy()}
    for k, e in embs.items():
        g6[f"emb{k}"] = e.numpy()
    for metric, norm in (("dot", False), ("cosine", True)):
        for align in ("softmax", "hardmax"):
            for temp in (1.0, 10.0):
                gm = begh.GroundingModule(D6, K6, 5, local_metric=metric, alignment=align, temperature=temp,
                                          normalize_emb=norm)
                gm.set_class_embeddings({k: v.clone() for k, v in embs.items()}, "cpu")
                x = lm.normalize_vec(img, dim=1) if norm else img       # the predictor normalises before the module (:423-424)
                scores, att = gm(x)
                tag = f"{metric}_{align}_t{int(temp)}"
                g6[tag + "_scores"], g6[tag + "_att"] = scores.detach().numpy(), att.detach().numpy()
                scores2, _ = gm(x)                                       # second call (num_tok was mutated in place)
                assert torch.equal(scores, scores2)
    np.savez_compressed(os.path.join(OUT, "g6_grounding_module.npz"), **g6)
    make_g7()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
