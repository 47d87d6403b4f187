#!/usr/bin/env python
"""bench.py -- proposals/sec through the LSM ROI head on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: starts N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic input resident in HBM, THROUGH THE PLUGIN SURFACE the
reference defines the metric on: `--images` res4 feature maps [B,1024,50,84] (1333x800 images, stride 16)
and `--proposals` boxes per image as Detectron2-style `Instances` / `Boxes` ->

    EmbeddingProposalsRes5ROIHeads._shared_roi_transform(features, boxes)      roi_emb_heads.py:243-245  (ROIAlign + Res5)
        .mean(dim=[2,3])                                                         :356  (fused into the call: pooled=True)
    EmbeddingFastRCNNOutputLayers.forward(box_features)                        box_emb_head.py:179-212 (bbox_pred, emb_pred,
                                                                                 (norm), similarity GEMM x (classes+1) x dim bank)

Images shard over ranks with no data-path collective (inference needs none): weak scaling.
`value` is scope S2 = the full head as the reference executes it (Res5 included).  Scope S1 = the north-star kernel list only
(ROIAlign + mean/FCs/similarity on a stand-in for the Res5 output) and the 1024-d bank variants are reported in "scopes".

`scopes.eval_1img` is the reference's EVALUATION call (TEST.IMS_PER_BATCH 1: one image x 1000 proposals per call through
roi_heads(images, features, proposals, None) incl. softmax / box decoding / score threshold / class-wise NMS / top-100), per image.
The default run's `train` object times one LSM and one STT training step of the path (fixed batch, then `ms_per_step_multiscale`:
a different batch of the reference's real training shapes every step) and, under N ranks or through a one-rank probe, places
DistributedDataParallel's buckets on the Res5 backward's timeline (`train.gradient_exchange.schedule`).

--mode train adds a `train` object: one LSM training step of the path per iteration (configs/coco_lsm.yaml: 4 images per GPU,
200 sampled proposals per image) -- EmbeddingProposalsRes5ROIHeads.forward with targets (roi_emb_heads.py:311-349: labelling /
sampling, whole-grid Res5, ROIAlign + Res5 + mean, box predictor, losses) + GroundingHead on the box branch
(grounding_head.py:92-388) + backward + SGD step, on the hand-written kernels and, beside it, with RES5_BACKEND miopen.
One JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3    # f32-input MFMA peak
MFMA_16BIT_PEAK_TFLOPS = 2500.0 # dense f16 / bf16 MFMA peak
SUSTAINED_F16_MFMA_TFLOPS = 1975.0          # measured, not nominal: profiles/r03_mfma_shape_probe.txt (reported beside `frac`, never instead of it)


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--mode", choices=["infer", "train"], default="infer",
                   help="train: the detailed training report (one --train-config, hip next to the MIOpen-autograd form of the same "
                        "module).  The default run already carries a compact `train` object: configs 4 and 5 on the hip backend")
    p.add_argument("--skip-train", action="store_true", help="no `train` object in the default run (profile runs)")
    p.add_argument("--images", type=int, default=8, help="images per GPU per step (SURVEY.md 8d: N = 8 / GPU)")
    p.add_argument("--proposals", type=int, default=1000)
    p.add_argument("--classes", type=int, default=1203)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--sim-dtype", choices=["fp32", "bf16"], default="fp32")
    p.add_argument("--res5", choices=["miopen", "hip"], default="hip")
    p.add_argument("--conv3x3", choices=["winograd", "direct"], default="winograd",
                   help="form of the Res5 3x3 convolutions on the hip backend")
    p.add_argument("--block0", choices=["map", "pooled"], default="map",
                   help="run Res5 block 0's 1x1 convolutions on the map (before ROIAlign) or on the pooled rows")
    p.add_argument("--res5-dtype", choices=["fp32", "f16x2", "bf16"], default="f16x2",
                   help="arithmetic of the Res5 GEMMs: f16x2 = fp32 in / fp32 out with the products formed from split "
                        "(hi, lo) f16 operand pairs on the f16 matrix pipe, fp32 accumulate (fp32-level accuracy, "
                        "csrc/gemm_split.hip); fp32 = the f32 MFMA; bf16 = reduced-precision operands (never a headline)")
    p.add_argument("--train-images", type=int, default=4, help="--mode train: images per GPU (IMS_PER_BATCH 32 / 8 GPUs)")
    p.add_argument("--train-samples", type=int, default=200, help="--mode train: ROI_HEADS.BATCH_SIZE_PER_IMAGE (coco_lsm.yaml:32)")
    p.add_argument("--train-config", choices=["lsm", "stt"], default="lsm",
                   help="--mode train: lsm = configs/coco_lsm.yaml step (EmbeddingProposalsRes5ROIHeads + GroundingHead, 4 img x 200 "
                        "sampled); stt = configs/coco_stt.yaml fine-tune step (EmbeddingRes5ROIHeads, 48 classes, emb_pred frozen, "
                        "3 img x 512 sampled)")
    p.add_argument("--train-backends", default="hip,miopen", help="--mode train: which Res5 backends to time (profile runs: hip)")
    p.add_argument("--unfrozen-steps", type=int, default=-1,
                   help="training steps timed WITHOUT gc.freeze() for train.*.ms_per_step_unfrozen_heap (-1: max(3 x steps, 50); 0: skip)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--skip-f32-reference", action="store_true",
                   help="do not also time the f32-MFMA form of the Res5 GEMMs (profile runs)")
    p.add_argument("--skip-s1", action="store_true",
                   help="only the S2 scope (profiling runs: the kernel mix then equals the timed region's)")
    p.add_argument("--skip-variants", action="store_true", help="do not time the 1024-d bank variants (profile runs)")
    p.add_argument("--skip-eval", action="store_true", help="no scopes.eval_1img (the reference's evaluation call incl. post-processing)")
    p.add_argument("--only-eval", action="store_true",
                   help="profile runs: ONLY the scopes.eval_1img loop (the kernel mix of the process is then the evaluation call's)")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline wall time")
    # developer/test knobs: rehearse the multi-process flow on a box with fewer GPUs than ranks
    p.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl")
    p.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (single-GPU test boxes; implies --dist-backend gloo)")
    p.add_argument("--force-dist", action="store_true",
                   help="initialise the process group and wrap the training step in DistributedDataParallel even with ONE rank: the RCCL "
                        "rehearsal a 1-GPU box allows (communicator, DDP's bucketed all-reduce hooks, the timing collectives)")
    p.add_argument("--ddp-bucket-mb", type=int, default=0,
                   help="DistributedDataParallel's bucket_cap_mb for the training steps; 0 = chosen per configuration so that a bucket closes "
                        "with every bottleneck's conv2 + conv3 node: 17 MiB for LSM (emb_pred's 6 MiB arrive first: 6 + 9 + 4 closes the "
                        "first bucket), 13 MiB = conv2 + conv3 for STT (emb_pred frozen).  torch's default is 25")
    p.add_argument("--multiscale-batches", type=int, default=16,
                   help="training: batches of the multi-scale pool behind train.*.ms_per_step_multiscale (0: skip)")
    p.add_argument("--skip-exchange-probe", action="store_true",
                   help="no train.gradient_exchange.schedule (the traced DDP steps that place the buckets on the Res5 backward's timeline)")
    args = p.parse_args(argv)
    if args.share_gpu and args.dist_backend == "nccl":
        # RCCL cannot place two ranks on one device (the attempt hangs for minutes before it fails): single-GPU test boxes use gloo
        print("bench.py: --share-gpu implies --dist-backend gloo", file=sys.stderr)
        args.dist_backend = "gloo"
    return args


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD `torch.distributed.run` (nothing in this
    process has touched the GPU yet, and it never will), relay rank 0's JSON line and the child's exit code."""
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if r.returncode != 0 or line is None:
        sys.stderr.write(r.stdout[-4000:])
        sys.stderr.write(f"\nbench.py: the {args.gpus}-rank child exited with code {r.returncode}\n")
        return r.returncode or 1
    print(line)
    return 0


def synth_boxes(gen, n: int):
    """SURVEY.md 8d boxes: centre uniform, log2(side) U[4, log2 800], aspect U[0.5,2], clipped; XYXY fp32 [n,4]."""
    import torch
    cx = torch.rand(n, generator=gen) * 1333.0
    cy = torch.rand(n, generator=gen) * 800.0
    side = 2.0 ** (4.0 + torch.rand(n, generator=gen) * (np.log2(800.0) - 4.0))
    aspect = 0.5 + 1.5 * torch.rand(n, generator=gen)
    w, h = side * aspect.sqrt(), side / aspect.sqrt()
    return torch.stack([(cx - w / 2).clamp(0, 1333), (cy - h / 2).clamp(0, 800),
                        (cx + w / 2).clamp(0, 1333), (cy + h / 2).clamp(0, 800)], dim=1).to(torch.float32)


MIN_SIZE_TRAIN = (640, 672, 704, 736, 768, 800)      # configs/coco_stt.yaml:54 (coco_lsm.yaml inherits the same Detectron2 default)
MAX_SIZE_TRAIN = 1333


def synth_image_size(gen):
    """(H, W) of one training image behind Detectron2's ResizeShortestEdge(MIN_SIZE_TRAIN, MAX_SIZE_TRAIN, "choice") [D2-upstream]: the
    short side drawn from MIN_SIZE_TRAIN, the long side from COCO-like aspect ratios (4:3 and 3:2 dominate; a quarter of the images
    are portrait), both scaled down when the long side would pass MAX_SIZE_TRAIN."""
    import torch
    short = MIN_SIZE_TRAIN[int(torch.randint(0, len(MIN_SIZE_TRAIN), (1,), generator=gen))]
    ar = (4 / 3, 4 / 3, 4 / 3, 3 / 2, 3 / 2, 16 / 9, 1.0, 5 / 4)[int(torch.randint(0, 8, (1,), generator=gen))]
    long = short * ar
    if long > MAX_SIZE_TRAIN:
        short, long = short * MAX_SIZE_TRAIN / long, MAX_SIZE_TRAIN
    h, w = int(short + 0.5), int(long + 0.5)
    if float(torch.rand(1, generator=gen)) < 0.25:
        h, w = w, h
    return h, w


def synth_boxes_in(gen, n: int, h: int, w: int):
    """synth_boxes for an image of h x w pixels (side up to the image's short side)."""
    import torch
    cx = torch.rand(n, generator=gen) * w
    cy = torch.rand(n, generator=gen) * h
    side = 2.0 ** (4.0 + torch.rand(n, generator=gen) * (np.log2(float(min(h, w))) - 4.0))
    aspect = 0.5 + 1.5 * torch.rand(n, generator=gen)
    bw, bh = side * aspect.sqrt(), side / aspect.sqrt()
    return torch.stack([(cx - bw / 2).clamp(0, w), (cy - bh / 2).clamp(0, h),
                        (cx + bw / 2).clamp(0, w), (cy + bh / 2).clamp(0, h)], dim=1).to(torch.float32)


def synth_train_batch(gen, device, n_images, n_props, n_classes, dim, *, multiscale=False, max_gt=7, fixed_gt=True, short_image=None,
                      crowded_image=None):
    """One rank's synthetic training batch: res4 maps, RPN-style proposals (a few of them near the ground truth), targets, captions.
    multiscale: every image its own size (synth_image_size), the batch padded to its largest image as Detectron2's ImageList does
    [D2-upstream] -> res4 map [B, 1024, ceil(Hmax / 16), ceil(Wmax / 16)]; else the fixed 1333 x 800 batch.
    fixed_gt: max_gt boxes on every image; else 0..max_gt (images WITHOUT ground truth included).
    short_image: (index, n) -- that image gets only n proposals (fewer candidates than the sampling budget: the labelling cannot
    speculate, the forward waits for the true counts).
    crowded_image: index -- every proposal of that image sits on a ground-truth box (all foreground): with a POSITIVE_FRACTION < 1
    (not the reference's: both configurations use 1.0) the background candidates do not fill the rest of the budget and the
    speculated sample MISSES."""
    import torch
    from locov_amd.structures import Boxes, Instances
    sizes = [synth_image_size(gen) if multiscale else (800, 1333) for _ in range(n_images)]
    Hm, Wm = max(h for h, _ in sizes), max(w for _, w in sizes)
    mh, mw = -(-Hm // 16), -(-Wm // 16)
    features = torch.randn(n_images, 1024, mh, mw, generator=gen).to(device)
    proposals, targets = [], []
    for i, (h, w) in enumerate(sizes):
        n_gt = max_gt if fixed_gt else int(torch.randint(0, max_gt + 1, (1,), generator=gen))
        if crowded_image == i:
            n_gt = max(n_gt, 1)
        R = n_props if short_image is None or short_image[0] != i else short_image[1]
        gt = synth_boxes_in(gen, n_gt, h, w) if multiscale else synth_boxes(gen, n_gt)
        b = synth_boxes_in(gen, R, h, w) if multiscale else synth_boxes(gen, R)
        k = min(n_gt, R)
        if crowded_image == i:
            b = gt[torch.arange(R) % n_gt] + torch.rand(R, 4, generator=gen) * 2 - 1
            b = b.clamp(min=0)
        elif k:
            b[:k] = (gt[:k] + torch.rand(k, 4, generator=gen) * 8 - 4).clamp(min=0)
        if multiscale:                                   # (RPN proposals are clipped to their image; the fixed batch keeps round 5's data)
            b[:, 0::2] = b[:, 0::2].clamp(max=float(w) - 1.0)
            b[:, 1::2] = b[:, 1::2].clamp(max=float(h) - 1.0)
        b[:, 2:] = torch.maximum(b[:, 2:], b[:, :2] + 1.0)
        p = Instances((h, w))
        p.proposal_boxes = Boxes(b.to(device))
        p.objectness_logits = torch.zeros(R, device=device)
        t = Instances((h, w))
        t.gt_boxes = Boxes(gt.to(device))
        t.gt_classes = torch.randint(0, n_classes, (n_gt,), generator=gen).to(device)
        proposals.append(p)
        targets.append(t)
    caption = {"input_embeddings": torch.randn(n_images, 70, dim, generator=gen).to(device),
               "attention_mask": torch.ones(n_images, 70, device=device),
               "special_tokens_mask": torch.zeros(n_images, 70, device=device)}
    caption["special_tokens_mask"][:, 0] = 1
    return {"features": features, "proposals": proposals, "targets": targets, "caption": caption, "sizes": sizes, "map": (mh, mw)}


def build_heads(args, device, *, dim=None, sim_dtype=None, res5=None, res5_dtype=None, train=False, seed=1992, stt=False):
    """The plugin exactly as train_ovnet.py gets it: cfg -> build_roi_heads (roi_emb_heads.py:168-214) -> load_embeddings'
    set_class_embeddings (trainer.py:365-396).  Random-init weights (He-init Res5, FrozenBN identity, emb_pred N(0, 0.01))."""
    import torch
    import locov_amd
    from locov_amd.structures import ShapeSpec
    cfg = locov_amd.config.get_cfg()
    cfg.MODEL.ROI_HEADS.NAME = "EmbeddingProposalsRes5ROIHeads"
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = args.classes
    cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = args.train_samples          # coco_lsm.yaml:32
    cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION = 1.0                            # coco_lsm.yaml:30
    cfg.MODEL.ROI_HEADS.DETACH_CLASS_PREDICTOR = True                      # coco_lsm.yaml:31
    cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED = True
    cfg.MODEL.ROI_BOX_HEAD.FREEZE_EMB_PRED = False
    cfg.MODEL.ROI_BOX_HEAD.EMB_DIM = dim or args.dim
    cfg.MODEL.ROI_BOX_HEAD.SIM_GEMM_DTYPE = sim_dtype or args.sim_dtype
    cfg.MODEL.ROI_BOX_HEAD.RES5_BACKEND = res5 or args.res5
    cfg.MODEL.ROI_BOX_HEAD.RES5_CONV3X3 = args.conv3x3
    cfg.MODEL.ROI_BOX_HEAD.RES5_DTYPE = res5_dtype or args.res5_dtype
    classes = args.classes
    if stt:                                                                # configs/coco_stt.yaml:17-37
        cfg.MODEL.ROI_HEADS.NAME = "EmbeddingRes5ROIHeads"
        cfg.MODEL.ROI_HEADS.NUM_CLASSES = classes = 48
        cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 512                     # Detectron2's default (not overridden)
        cfg.MODEL.ROI_HEADS.DETACH_CLASS_PREDICTOR = False
        cfg.MODEL.ROI_BOX_HEAD.FREEZE_EMB_PRED = True
    torch.manual_seed(seed)                                                # configs/coco_lsm.yaml:126
    heads = locov_amd.build_roi_heads(cfg, {"res4": ShapeSpec(channels=1024, stride=16)}).to(device)
    heads.train(train)
    gen = torch.Generator().manual_seed(seed + 1)
    bank = torch.randn(classes + 1, cfg.MODEL.ROI_BOX_HEAD.EMB_DIM, generator=gen) * 0.05
    bank[-1] = 0
    heads.box_predictor.set_class_embeddings(bank)
    heads.num_classes = heads.box_predictor.num_classes
    if args.block0 == "pooled":                                            # bench-only knob: never take the map path
        heads.res5.map_path_pays = lambda *a, **k: False
    return heads, cfg


class Workload:
    def __init__(self, args, device):
        import torch
        from locov_amd import ops
        from locov_amd.structures import Boxes, Instances
        self.ops, self.args, self.device = ops, args, device
        gen = torch.Generator().manual_seed(1992)
        B, R = args.images, args.proposals
        self.features = {"res4": torch.randn(B, 1024, 50, 84, generator=gen).to(device)}
        self.proposals = []
        for _ in range(B):
            inst = Instances((800, 1333))
            inst.proposal_boxes = Boxes(synth_boxes(gen, R).to(device))
            inst.objectness_logits = torch.zeros(R, device=device)
            self.proposals.append(inst)
        self.boxes = [p.proposal_boxes for p in self.proposals]
        self.heads, _ = build_heads(args, device)
        self.r5_standin = torch.randn(B * R, 2048, 7, 7, generator=gen).clamp_(min=0).to(device)
        rois = torch.cat([torch.cat([torch.full((R, 1), float(i)), p.proposal_boxes.tensor.cpu()], 1)
                          for i, p in enumerate(self.proposals)]).to(device)
        self.rois = rois
        self.ev = []        # (start, end) events around the dominant hand-written kernel in miopen mode (ROIAlign)

    def step_s2(self, heads=None, timed=False):
        """_shared_roi_transform -> mean -> box_predictor.forward, through the plugin."""
        import torch
        heads = heads or self.heads
        with torch.no_grad():
            feats = [self.features[f] for f in heads.in_features]
            if timed and heads.res5_backend != "hip":
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                x = heads.pooler(feats, self.boxes)
                e1.record()
                self.ev.append((e0, e1))
                box_features = heads._pooled_mean(heads.res5(x))
            else:
                box_features = heads._shared_roi_transform(feats, self.boxes, pooled=True)      # :355-356
            return heads.box_predictor(box_features)                                             # :357 (scores, deltas)

    def eval_heads(self, sigma: float = 3.0):
        """The heads of the evaluation scope: the SAME Res5 / predictor weights with a bank scaled so that the class logits have
        standard deviation `sigma` over the workload's proposals.  Random-init logits are ~1e-2, their softmax is uniform and NOTHING
        passes SCORE_THRESH_TEST 0.05 -- the post-processing would be timed on zero candidates.  sigma = 3 over 1203 classes puts
        ~2-3 (proposal, class) pairs per proposal above the threshold, i.e. a few thousand NMS candidates per image (what a trained
        open-vocabulary head yields); swapping the bank is what trainer.py:187-191 does per test set."""
        import torch
        hit = self.__dict__.get("_eval_heads")
        if hit is not None:
            return hit
        h, _ = build_heads(self.args, self.device)
        h.res5 = self.heads.res5
        h.box_predictor.load_state_dict(self.heads.box_predictor.state_dict())
        with torch.no_grad():
            feats = [self.features["res4"][:1]]
            scores, _ = h.box_predictor(h._shared_roi_transform(feats, self.boxes[:1], pooled=True))
            std = float(scores[:, :-1].std())
            bank = self.heads.box_predictor.cls_score.weight.detach().clone() * (sigma / max(std, 1e-30))
            h.box_predictor.set_class_embeddings(bank)
        self._eval_heads = h
        self.eval_logit_sigma = sigma
        return h

    def step_eval(self, n_images: int = 1):
        """The reference's EVALUATION call (configs/coco_stt.yaml:50, coco_lsm.yaml:121: TEST.IMS_PER_BATCH 1 -> one image x
        POST_NMS_TOPK_TEST 1000 proposals per call): roi_heads(images, features, proposals, None) -> inference_detection
        (roi_emb_heads.py:351-360): ROIAlign + Res5 + mean, the box predictor AND box_predictor.inference (softmax, box
        decoding, score threshold, class-wise NMS, top-100)."""
        import torch
        h = self.eval_heads()
        with torch.no_grad():
            feats = {"res4": self.features["res4"][:n_images]}
            return h(None, feats, self.proposals[:n_images], None)

    def step_s1(self):
        import torch
        with torch.no_grad():
            self.ops.roi_align(self.features["res4"], self.rois, 14, 1.0 / 16, 0, True)
            return self.heads.box_predictor(self.heads._pooled_mean(self.r5_standin))


class TrainWorkload:
    """One LSM training step of the path (see the module docstring)."""

    def __init__(self, args, device, backend, world, data_seed=1992, config=None, ddp=None):
        import torch
        import locov_amd
        from locov_amd.grounding_head import GroundingHead
        self.args, self.device = args, device
        self.stt = (config or args.train_config) == "stt"
        self.n_images = 3 if self.stt else args.train_images                 # IMS_PER_BATCH 24 / 8 GPUs (coco_stt.yaml:41)
        self.n_classes = 48 if self.stt else args.classes
        ddp = (world > 1 or args.force_dist) if ddp is None else bool(ddp)
        self.bucket_cap_mb = args.ddp_bucket_mb or (13 if self.stt else 17)
        self.set_data(data_seed)
        self.heads, cfg = build_heads(args, device, res5=backend, train=True, stt=self.stt)
        if self.stt:
            heads = self.heads

            class STTStep(torch.nn.Module):
                """OvrRCNN's training forward from the ROI heads on (ovr_rcnn.py:28-74): losses of EmbeddingRes5ROIHeads."""

                def __init__(self):
                    super().__init__()
                    self.heads = heads

                def forward(self, feat, proposals, targets, caption):
                    _, losses = self.heads(None, {"res4": feat}, proposals, targets)
                    return sum(losses.values()), self.heads.batch_size_per_image * len(proposals)

            self.module = STTStep()
            self.run = self.module
            if ddp:
                from torch.nn.parallel import DistributedDataParallel as DDP
                self.run = DDP(self.module, device_ids=None if args.share_gpu else [device.index], broadcast_buffers=False,
                               bucket_cap_mb=self.bucket_cap_mb)
            params = [p for p in self.module.parameters() if p.requires_grad]
            self.opt = torch.optim.SGD(params, lr=0.005, momentum=0.9, weight_decay=1e-4)   # coco_stt.yaml:42
            return
        cfg.MODEL.MMSS_HEAD.DISTILLATION_LOSS = False
        cfg.MODEL.MMSS_HEAD.GROUNDING.LOSS = "cross_entropy"
        cfg.MODEL.MMSS_HEAD.GROUNDING.ALIGN_WORDS_TO_REGIONS = True
        cfg.MODEL.MMSS_HEAD.GROUNDING.ALIGN_REGIONS_TO_WORDS = True
        self.grounding = GroundingHead(cfg, 2048, args.dim).to(device)
        # weight tying of distill_prop_mmss_gcnn.py:117-125: emb_pred IS the grounding head's v2l_projection
        self.heads.box_predictor.emb_pred.weight = self.grounding.v2l_projection.weight
        self.heads.box_predictor.emb_pred.bias = self.grounding.v2l_projection.bias
        n_regions = 100                                                                  # MMSS_HEAD.SPATIAL_DROPOUT
        heads, grounding, dev_ = self.heads, self.grounding, device

        class LSMStep(torch.nn.Module):
            """forward of one training step of the path -> (loss, sampled proposals): what DDP wraps."""

            def __init__(self):
                super().__init__()
                self.heads, self.grounding = heads, grounding

            def forward(self, feat, proposals, targets, caption):
                grid, box_feats, sampled, losses = self.heads(None, {"res4": feat}, proposals, targets)
                n = min(n_regions, min(f.shape[0] for f in box_feats))
                regions = torch.stack([f[:n] for f in box_feats])           # [B, n, 2048]  (distill_prop_mmss_gcnn.py:348-417)
                image = {"region_features": regions, "region_mask": torch.ones(regions.shape[:2], device=dev_)}
                _, g_losses = self.grounding(image, caption)
                # (the whole-grid features feed the grid grounding branch in the reference; a mean keeps their backward in the step)
                return sum(losses.values()) + sum(g_losses.values()) + grid.mean(), sum(len(s) for s in sampled)

        self.module = LSMStep()
        self.run = self.module
        if ddp:       # the gradient exchange of the path: DDP's bucketed all-reduce (RCCL over xGMI), overlapped with backward
            from torch.nn.parallel import DistributedDataParallel as DDP
            self.run = DDP(self.module, device_ids=None if args.share_gpu else [device.index], broadcast_buffers=False,
                           bucket_cap_mb=self.bucket_cap_mb)
        params = [p for p in self.module.parameters() if p.requires_grad]
        self.opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, weight_decay=1e-4)   # coco_lsm.yaml:104-105

    def set_data(self, seed):
        """The FIXED synthetic batch of one rank (1333 x 800 images, args.proposals proposals, seven GT boxes per image)."""
        import torch
        gen = torch.Generator().manual_seed(seed)
        self.use_batch(synth_train_batch(gen, self.device, self.n_images, self.args.proposals, self.n_classes, self.args.dim))

    def use_batch(self, batch):
        self.features, self.proposals, self.targets, self.caption = batch["features"], batch["proposals"], batch["targets"], batch["caption"]

    def make_pool(self, n_batches: int, seed: int, n_props: int = 2000, underfill_every: int = 0):
        """`n_batches` batches of the reference's REAL training shapes, resident in HBM: 2 000 RPN proposals per
        image into the labelling (Detectron2's POST_NMS_TOPK_TRAIN default), every image its own size out of
        configs/coco_stt.yaml:54's MIN_SIZE_TRAIN (the batch padded to its largest image), 0-15 GT boxes per image incl. none, and
        every `underfill_every`-th batch with one image of fewer proposals than the sampling budget: the labelling cannot speculate
        and the forward waits for the true counts.  (Both reference configurations sample with POSITIVE_FRACTION 1.0 --
        coco_lsm.yaml:30, coco_stt.yaml:25 -- under which ANY `budget` candidates fill the budget: with no ignore band in the
        matcher a speculated sample cannot miss; the miss leg is driven by tests with an ignore-band matcher.)"""
        import torch
        gen = torch.Generator().manual_seed(seed)
        budget = self.heads.batch_size_per_image
        self.pool = []
        for i in range(n_batches):
            under = bool(underfill_every) and i % underfill_every == underfill_every - 1
            short = (i % self.n_images, max(budget // 2, 8)) if under else None
            self.pool.append(synth_train_batch(gen, self.device, self.n_images, n_props, self.n_classes, self.args.dim,
                                               multiscale=True, max_gt=15, fixed_gt=False, short_image=short))
        self.pool_at = 0
        return self.pool

    def step_pool(self):
        """One training step on the NEXT batch of the pool (a different map size, proposal set and ground truth every step)."""
        self.use_batch(self.pool[self.pool_at % len(self.pool)])
        self.pool_at += 1
        return self.step()

    def forward_backward(self, scale: float = 1.0):
        """Forward + backward of one step (gradients ACCUMULATE into .grad; under DDP they are averaged over the ranks)."""
        feat = self.features.detach().requires_grad_(True)          # the backbone trains (FREEZE_AT 0): res4 needs its gradient
        loss, n_sampled = self.run(feat, self.proposals, self.targets, self.caption)
        (loss * scale).backward()
        return loss.detach(), n_sampled

    def step(self):
        self.opt.zero_grad(set_to_none=True)
        _, n_sampled = self.forward_backward()
        self.opt.step()
        return n_sampled

    def trace_exchange(self, steps: int = 3):
        """Where DDP's buckets become ready inside the Res5 backward (locov_amd.sharding.GradientExchangeTrace), from the LAST of
        `steps` traced training steps of this (DDP-wrapped) workload."""
        from locov_amd.sharding import GradientExchangeTrace
        if self.run is self.module:
            raise RuntimeError("trace_exchange: the workload is not wrapped in DistributedDataParallel")
        trace = self.__dict__.get("_trace")
        if trace is None:
            trace = self._trace = GradientExchangeTrace(self.run, self.module.named_parameters(), self.device)
        rep = None
        for _ in range(steps):
            self.opt.zero_grad(set_to_none=True)
            trace.start()
            self.forward_backward()
            rep = trace.stop()
            self.opt.step()
        return rep


TRAFFIC_FILE = "r06_pmc_traffic.json"


def traffic_record(args):
    """(record, reason): the committed PMC passes (profiles/<TRAFFIC_FILE>: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of
    this same command on ANOTHER box, corrected as MI355X_MICROARCH.md prescribes) when they describe THIS run -- the same workload
    and a library built from the same kernel sources -- else (None, why not).  Counters cannot be collected from inside the timed
    run; a record that does not match is refused loudly (roofline.traffic null + roofline.traffic_source says why), never used
    silently."""
    path = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    try:
        with open(path) as f:
            rec = json.load(f)
        wl = rec["workload"]
    except (OSError, KeyError, ValueError) as e:
        return None, f"profiles/{TRAFFIC_FILE} unreadable ({type(e).__name__})"
    diff = [k for k in ("images", "proposals", "classes", "dim", "res5", "conv3x3", "block0", "res5_dtype") if wl.get(k) != getattr(args, k)]
    if diff:
        return None, f"profiles/{TRAFFIC_FILE} was taken on another workload (differs in {', '.join(diff)})"
    # staleness: the PMC pass names the kernel sources it was taken on (locov_amd.build.source_fingerprint at collection
    # time); a library built from other sources moves different bytes, so the recorded figure is refused
    from locov_amd import build as _build
    have, want = rec.get("source_fingerprint"), _build.source_fingerprint()
    if have != want:
        return None, (f"profiles/{TRAFFIC_FILE} is STALE: taken on kernel sources {have}, this library is built from {want} "
                      "(re-run tools/profile_round.sh)")
    return rec, None


def recorded_traffic(args, kernel_key: str):
    """HBM bytes per launch of the kernel(s) whose profiler name contains kernel_key, from traffic_record(); None without one."""
    rec, _ = traffic_record(args)
    if rec is None:
        return None
    try:
        keys = (kernel_key,) if isinstance(kernel_key, str) else tuple(kernel_key)
        hits = [v for name, v in rec["kernels"].items() if any(k in name for k in keys)]      # every matching template instance
        n = sum(v["launches_sampled"] for v in hits)
        if n:
            return sum(v["hbm_bytes_per_launch"] * v["launches_sampled"] for v in hits) / n
    except (KeyError, ValueError, TypeError):
        pass
    return None


def traffic_source(args) -> str:
    rec, why = traffic_record(args)
    if rec is None:
        return f"none -- {why}"
    return (f"profiles/{TRAFFIC_FILE}: rocprofv3 --pmc passes of this command on the same kernel sources (fingerprint "
            f"{rec.get('source_fingerprint')}), collected by the builder on ANOTHER box, not in this run")


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the whole host and oversubscribes a quota-limited container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def cpu_model() -> str:
    """CPU model string of this host (platform.processor() is empty on Linux)."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or platform.machine()


def cpu_baseline(args, seconds: float):
    """The oracle (a port: the reference's Python cannot run here) timed on this host's cores over a bounded
    sample of the same workload (forward only, median of >= 5 runs, the 1000-proposal shape and config 1's
    2 img x 100 proposals x 80 classes; core count and CPU model stated)."""
    import torch
    from oracle import lsm_oracle as oracle
    oracle.build()
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    os.environ["OMP_NUM_THREADS"] = str(ncores)
    rng = np.random.default_rng(1992)
    feat = rng.standard_normal((2, 1024, 50, 84)).astype(np.float32)
    params = oracle.make_res5_params(0)
    head = oracle.synth_head(rng, 2048, args.dim, args.classes)
    head80 = oracle.synth_head(rng, 2048, args.dim, 80)

    def run(n, images=1, h=head):
        boxes = [oracle.synth_boxes(rng, n) for _ in range(images)]
        t0 = time.perf_counter()
        oracle.roi_head_forward(feat[:images], boxes, params, h)
        return time.perf_counter() - t0

    run(8)                                   # warm-up (thread pools, page-in)
    t_probe = run(50)
    runs = 5
    # ~2/3 of the budget on the 1000-proposal shape (a bounded sample of one image's proposals), the rest on config 1
    n = int(max(50, min(args.proposals, 50 * (seconds * 0.66 / runs) / max(t_probe, 1e-3))))
    t = float(np.median([run(n) for _ in range(runs)]))
    t1 = float(np.median([run(100, images=2, h=head80) for _ in range(runs)]))
    model = cpu_model()
    return {"value": n / t, "unit": "proposals/s", "cores": ncores, "kind": "port", "cpu_model": model,
            "sample": f"{n} proposals of one 1333x800 image, full head (ROIAlign+Res5+mean+FCs+sim K={args.classes}), "
                      f"median of {runs} runs, {t:.2f} s each; torch CPU convs + OpenMP C oracle, {ncores} cores of {model}",
            "config1": {"value": 200 / t1, "unit": "proposals/s",
                        "sample": f"BASELINE config 1: 2 images x 100 proposals, 80-class bank, median of {runs} runs, {t1:.2f} s each"}}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))            # before anything initialises the GPU in this process
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the hot path has no CPU fallback)")
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    rccl_ranks = 1
    dist_on = world > 1 or args.force_dist
    if dist_on:
        if world == 1:                        # --force-dist without a launcher: a one-rank rendezvous on the loopback
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":       # RCCL over xGMI
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
        ones = torch.ones(1, device=device if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(ones)                 # the collective the training step relies on, exercised explicitly
        rccl_ranks = int(ones.item())
        if rccl_ranks != world:
            raise SystemExit(f"all_reduce over {world} ranks returned {rccl_ranks}")

    from locov_amd import _lib
    from locov_amd.sharding import all_ranks, max_over_ranks
    lib = _lib.load()
    wl = Workload(args, device)

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    wall = {}
    per_rank = {}                               # key -> every rank's ms per step of that bracket (rank order)
    step_stats = {}                             # key -> median / min / max of the single steps of that bracket on this rank (ms)

    def timed(fn, steps, warmup, on_start=None, key=None, freeze=True, **kw):
        """K steps between barrier + synchronize on both sides; the time is hipEventElapsedTime between two events recorded
        on the launch stream inside that bracket, max over ranks; the host wall clock of the same bracket is
        kept beside it (wall[key]).  The interpreter's full garbage-collection pass is taken HERE, in front of the warm-up steps and
        the bracket, and the objects alive at that point are frozen (gc.freeze): with a quarter of a million tracked objects a
        generation-2 pass takes ~90 ms, and one landing at a random step inside a 10-step training window moved
        `train.lsm.ms_per_step` between 14.5 and 24 ms from run to run (tools/attic/ab_fused_losses.py).  Collections of the younger
        generations keep running inside the bracket, and the heap is un-frozen behind it (a frozen object is never collected:
        freezing per phase would pin every earlier workload's tensors).  freeze=False: nothing of that -- the interpreter's
        collector runs as it does under an unchanged train_ovnet.py (`train.*.ms_per_step_unfrozen_heap`)."""
        import gc
        # (the full pass first, the warm-up steps behind it: a full pass over an un-frozen heap keeps the host busy for ~0.1 s, long
        # enough for the idle GPU to drop its clocks -- a short bracket right behind it, e.g. the ten 2.5 ms ROIAlign calls of
        # S1_roi_align_roofline, then ran 10 % slower than the same kernel measures in steady state)
        if freeze:
            gc.collect()
            gc.freeze()
        for _ in range(warmup):
            fn()
        barrier()
        if on_start is not None:
            on_start()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps)] if key is not None else []
        t0 = time.perf_counter()
        e0.record()
        for i in range(steps):
            fn(**kw)
            if marks:
                marks[i].record()                   # (one event per step: the distribution of the single steps, beside the bracket's mean)
        e1.record()
        barrier()
        dt_wall = time.perf_counter() - t0
        if marks:
            ends = [e0] + marks
            single = sorted(ends[i].elapsed_time(ends[i + 1]) for i in range(steps))
            step_stats[key] = {"median": single[len(single) // 2], "min": single[0], "max": single[-1]}
        if freeze:
            gc.unfreeze()
        dt = e0.elapsed_time(e1) * 1e-3
        if key is not None:
            wall[key] = max_over_ranks(dt_wall, device if args.dist_backend == "nccl" else None)
            per_rank[key] = [t / steps * 1e3 for t in all_ranks(dt, device if args.dist_backend == "nccl" else None)]
        # the slowest rank defines the job's time
        return max_over_ranks(dt, device if args.dist_backend == "nccl" else None)

    def eval_scope(batched=True):
        """scopes.eval_1img: the reference's evaluation call (TEST.IMS_PER_BATCH 1) incl. the post-processing, per image.
        batched: also the same call on the bench's whole batch (per image); False for profile runs (--only-eval: the process's kernel
        mix is then the one-image call's)."""
        wl.eval_heads()
        steps_e = max(args.steps, 10)
        dte = timed(wl.step_eval, steps_e, 5, key="eval1")
        inst = wl.step_eval()[0][0]
        n_b = max(args.steps // 2, 3)
        dte8 = timed(lambda: wl.step_eval(args.images), n_b, 2) if batched else float("nan")
        ms1 = dte / steps_e * 1e3
        lib_launches = None
        try:
            c0 = int(lib.locov_launch_count())
            wl.step_eval()
            lib_launches = int(lib.locov_launch_count()) - c0
        except Exception:                 # noqa: BLE001
            pass
        return {"what": "roi_heads(images, features, proposals, None) -> inference_detection (roi_emb_heads.py:351-360) on ONE image x "
                        f"{args.proposals} proposals x {args.classes} classes: ROIAlign + Res5 + mean + predictor + softmax / box decoding / "
                        "score threshold 0.05 / class-wise NMS 0.5 / top-100 (configs/coco_stt.yaml:50 TEST.IMS_PER_BATCH 1)",
                "ms_per_image": ms1, "proposals_per_s": args.proposals * world / (ms1 * 1e-3),
                "single_calls_ms": step_stats["eval1"], "wall_ms_per_image": wall["eval1"] / steps_e * 1e3,
                "detections_per_image": len(inst), "logit_sigma": wl.eval_logit_sigma,
                "library_launches_per_image": lib_launches,
                "launches": "library_launches_per_image counts this library's launch checks in one call (live); every kernel of the call "
                            "incl. torch's: profiles/r06_eval_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py --only-eval`)",
                f"batched_{args.images}img_ms_per_image": dte8 / n_b / args.images * 1e3,
                "bank": "the bench's bank scaled to logit sigma 3 over the proposals (random-init logits pass no score threshold)"}

    if args.only_eval:
        if rank == 0:
            print(json.dumps({"eval_1img": eval_scope(batched=False)}))
        if dist_on:
            dist.destroy_process_group()
        return

    props_per_step = args.images * args.proposals * world
    # dominant kernel: the library brackets each of its GEMM-kernel launches with HIP events on the
    # launch stream while this is enabled (include/locov_hip.h, locov_gemm_timing_*)
    dt2 = timed(wl.step_s2, args.steps, args.warmup, on_start=lambda: lib.locov_gemm_timing_enable(1), key="s2", timed=True)

    def gemm_class(cls):
        import ctypes
        n, ms, fl, by = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _lib.check(lib.locov_gemm_timing_read_ex(cls, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)))
        return n.value, ms.value, fl.value, by.value
    gemm_plain, gemm_conv = gemm_class(0)[:3], gemm_class(1)[:3]
    gemm_plain_bf16, gemm_conv_bf16 = gemm_class(3)[:3], gemm_class(4)[:3]
    # the split-operand GEMM by launch kind (include/locov_hip.h, locov_gemm_timing_*): kind -> (timing class, kernel-name key of
    # the profiler output, what it is)
    SPLIT_KINDS = {
        "winograd_domain_batched": (10, "gemm_split_big_kernel<0, true>", "the 121 Winograd-domain GEMMs [R,512]x[512,512]^T of a 3x3 convolution, one launch"),
        "conv3": (9, "gemm_split_big_kernel<0, false>", "conv3 of blocks 0-1: [49R,512]x[2048,512]^T + residual, block output written in the split layout"),
        "conv3_mean_fused": (11, "gemm_split_big_kernel<1, false>", "conv3 of block 2 with the spatial mean in the epilogue (no [49R,2048] result)"),
        "conv1_with_winograd_input_transform": (8, "gemm_split_big_kernel<2, false>", "conv1 of blocks 1-2: [49R,2048]x[512,2048]^T, the epilogue writes "
                                                "the transform-domain tensor V instead of the pixels"),
        "tile_128x128": (5, "gemm_split_kernel<", "the launches below 1 024 tiles of 256x256: block 0's 1x1 convolutions on the map, emb_pred, cls_score"),
    }
    split_kinds = {k: gemm_class(c) for k, (c, _, _) in SPLIT_KINDS.items()}
    big_kinds = [k for k in SPLIT_KINDS if k != "tile_128x128"]
    gemm_split_big = tuple(sum(split_kinds[k][i] for k in big_kinds) for i in range(4))     # the dominant kernel: every 256x256 launch
    gemm_split = tuple(a + b for a, b in zip(gemm_split_big, split_kinds["tile_128x128"]))  # every split-operand GEMM launch of the step
    lib.locov_gemm_timing_enable(0)
    # the same job with the Res5 GEMMs on the f32 MFMA (reported beside the headline when that is the split path)
    dt2_f32 = None
    if args.res5 == "hip" and args.res5_dtype == "f16x2" and not args.skip_f32_reference:
        wl.heads.res5_dtype = "fp32"
        dt2_f32 = timed(wl.step_s2, args.steps, args.warmup)
        wl.heads.res5_dtype = "f16x2"
    dom_ms = float(np.mean([a.elapsed_time(b) for a, b in wl.ev])) if wl.ev else float("nan")
    dt1 = timed(wl.step_s1, args.steps, args.warmup) if not args.skip_s1 else float("nan")
    # S1's dominant kernel on its own: the pooler-contract ROIAlign ([R,1024,14,14] out of the NCHW map, incl. its channels-last copy)
    s1_roi = None
    if not args.skip_s1:
        with torch.no_grad():
            dtr = timed(lambda: wl.ops.roi_align(wl.features["res4"], wl.rois, 14, 1.0 / 16, 0, True), max(args.steps // 2, 3), 2)
        R_all = args.images * args.proposals
        roi_bytes = args.images * 1024 * 50 * 84 * 4 + R_all * 5 * 4 + R_all * 1024 * 14 * 14 * 4      # map + rois read, pooled written
        ms_roi = dtr / max(args.steps // 2, 3) * 1e3
        s1_roi = {"kernel": "roi_align_nhwc2nchw_kernel (+ nchw_to_nhwc_kernel): bit-exact pooler-contract ROIAlign", "bound": "hbm",
                  "ms_per_call": ms_roi, "algorithmic_bytes": roi_bytes, "achieved": roi_bytes / (ms_roi * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                  "unit": "GB/s", "frac": roi_bytes / (ms_roi * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "what": "write-dominated; proposals up to ~180 px take their taps from an LDS window (the transpose tile's bytes), larger ones "
                          "gather at the texture path's rate (torchvision's per-sample order: 9-16 samples per bin), not at HBM's: "
                          "docs/experiments.md R4.5, R4.7"}
    # north_star's 1024-d bank, fp32 and bf16 similarity GEMM (config 3): the same job with another predictor / bank
    variants = {}
    if not args.skip_variants:
        for key, (dim, simdt) in {"dim1024_fp32": (1024, "fp32"), "dim1024_bf16sim": (1024, "bf16")}.items():
            h2, _ = build_heads(args, device, dim=dim, sim_dtype=simdt)
            h2.res5 = wl.heads.res5                           # same Res5 weights (and packed operands)
            dtv = timed(lambda: wl.step_s2(heads=h2), max(args.steps // 2, 3), 2)
            variants[key + "_proposals_per_s"] = props_per_step * max(args.steps // 2, 3) / dtv
            del h2

    eval_1img = None
    if not args.skip_eval and args.res5 == "hip":
        eval_1img = eval_scope()
        eval_1img["vs_batched_S2_rate"] = eval_1img["proposals_per_s"] / (props_per_step * args.steps / dt2)
        wl.__dict__.pop("_eval_heads", None)

    def time_train(config, backend):
        tw = TrainWorkload(args, device, backend, world, data_seed=1992 + rank, config=config)
        n_sampled = tw.step()
        steps = max(args.steps, 3)
        dtt = timed(tw.step, steps, max(args.warmup, 8), key="train_" + config)      # (the first steps still grow the allocator's pools and pick the operand scales)
        # the same step with the interpreter's collector left alone (no collect / freeze in front of the bracket): what a trainer
        # that does not freeze its heap gets.  A full collection of torch's ~250 k tracked objects takes ~90 ms and comes every few
        # dozen steps, so this figure is taken over enough steps to hold its share of them
        steps_u = max(3 * steps, 50) if args.unfrozen_steps < 0 else args.unfrozen_steps
        out = {"sampled_proposals_per_s": n_sampled * world * steps / dtt, "ms_per_step": dtt / steps * 1e3,
               "single_steps_ms": step_stats["train_" + config],
               "per_rank_ms_per_step": per_rank["train_" + config], "sampled_proposals_per_step_per_gpu": n_sampled}
        if steps_u > 0:
            import gc
            full0 = gc.get_stats()[2]["collections"]
            dtu = timed(tw.step, steps_u, 0, freeze=False)
            out.update(ms_per_step_unfrozen_heap=dtu / steps_u * 1e3, unfrozen_heap_steps=steps_u,
                       unfrozen_heap_full_collections=gc.get_stats()[2]["collections"] - full0)
        if args.multiscale_batches > 0:
            # the same step on the reference's REAL training shapes (TrainWorkload.make_pool): a different map size, 2 000
            # proposals per image, 0-15 GT boxes and, every 8th batch, an image that cannot fill the sampling budget
            pool = tw.make_pool(args.multiscale_batches, seed=7 + rank, n_props=2000, underfill_every=8)
            tw.heads.stats.clear()
            steps_m = 2 * len(pool)
            dtm = timed(tw.step_pool, steps_m, len(pool), key="train_ms_" + config)       # (warm-up: every shape once)
            out.update(ms_per_step_multiscale=dtm / steps_m * 1e3,
                       multiscale={"steps": steps_m, "batches": len(pool), "proposals_per_image": 2000,
                                   "res4_maps": sorted({f"{b['map'][0]}x{b['map'][1]}" for b in pool}),
                                   "gt_boxes_per_image": "0-15", "underfilled_batches": sum(1 for i in range(len(pool)) if i % 8 == 7),
                                   "single_steps_ms": step_stats["train_ms_" + config],
                                   "what": "configs/coco_stt.yaml:54 MIN_SIZE_TRAIN (640 ... 800) x COCO-like aspect ratios, batch padded to "
                                           "its largest image, a different batch every step (pool resident in HBM)"},
                       speculation_misses=tw.heads.stats.get("speculation_misses", 0),
                       retry_stats=dict(tw.heads.stats))
            del pool
            tw.pool = []
        if dist_on and not args.skip_exchange_probe:
            # behind the timed brackets (they ran DDP's stock all-reduce): three traced steps of the same DDP module
            tw.set_data(1992 + rank)
            out["exchange_schedule"] = tw.trace_exchange(3)
        del tw
        torch.cuda.empty_cache()
        return out

    def exchange_probe():
        """The N = 1 run has no process group: place DDP's buckets on the Res5 backward's timeline with a ONE-rank gloo group
        (the module, the autograd nodes and DDP's bucket logic are the N-rank ones; only the collective's transport differs),
        behind every timed bracket.  Any failure is reported, never raised: this is a diagnostic beside the figures."""
        import datetime
        try:
            dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1,
                                    timeout=datetime.timedelta(seconds=60))
        except Exception as e:            # noqa: BLE001
            return {"error": f"one-rank gloo group: {type(e).__name__}: {e}"[:300]}
        try:
            tw = TrainWorkload(args, device, "hip", 1, data_seed=1992, config="lsm", ddp=True)
            for _ in range(3):
                tw.step()
            rep = tw.trace_exchange(3)
            rep["from"] = "one-rank gloo DistributedDataParallel of the LSM step, three traced steps behind the timed brackets"
            del tw
            return rep
        except Exception as e:            # noqa: BLE001
            return {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            dist.destroy_process_group()
            torch.cuda.empty_cache()

    def gradient_exchange(tr):
        caps = f"bucket_cap_mb {args.ddp_bucket_mb}" if args.ddp_bucket_mb else "bucket_cap_mb 17 (LSM) / 13 (STT)"
        ge = {"how": (f"DistributedDataParallel over {world} ranks ({args.dist_backend}), {caps}" if dist_on
                      else f"none (1 rank); under N ranks: DistributedDataParallel, {caps}"),
              "autograd_nodes": "Res5OutputFn + two Res5BlockFn per bottleneck (locov_amd/res5_train.py: tail = conv2 + conv3, head = conv1 + "
                                "shortcut): a half-block's weight gradients reach DDP's hooks when that half's backward kernels are enqueued; "
                                "which parameters DDP's (rebuilt) buckets hold and when they are ready: `schedule`",
              "bytes": "res5 59.8 MB + emb_pred 6.3 MB + bbox_pred 33 kB fp32 (SURVEY 8e)"}
        if args.skip_exchange_probe:
            return ge
        if dist_on:
            ge["schedule"] = {k: tr[k].pop("exchange_schedule", None) for k in ("lsm", "stt") if isinstance(tr.get(k), dict)}
        else:
            ge["schedule"] = {"lsm": exchange_probe()}
        return ge

    train = None
    if args.mode == "infer" and not args.skip_train and args.res5 == "hip" and args.res5_dtype in ("f16x2", "fp32"):
        # BASELINE configs 4 and 5 in the default line (compact: the hand-written path only; `--mode train` adds the
        # MIOpen-autograd form of the same module): one training step of the path per iteration, forward + backward + SGD,
        # wrapped in DistributedDataParallel (RCCL all-reduce of the path's gradients) when there is more than one rank
        train = {"lsm": dict(time_train("lsm", "hip"), config="configs/coco_lsm.yaml",
                             what=f"{args.train_images} img/GPU x {args.proposals} proposals -> {args.train_samples} sampled/img; "
                                  "EmbeddingProposalsRes5ROIHeads.forward(targets) (labelling, whole-grid Res5, ROIAlign + Res5 + mean, "
                                  "predictor, losses) + GroundingHead (box branch) + backward + SGD step"),
                 "stt": dict(time_train("stt", "hip"), config="configs/coco_stt.yaml",
                             what=f"3 img/GPU x {args.proposals} proposals -> 512 sampled/img, 48-class bank, emb_pred frozen; "
                                  "EmbeddingRes5ROIHeads.forward(targets) + backward + SGD step"),
                 "backend": "hip", "res5_dtype": args.res5_dtype}
        train["gradient_exchange"] = gradient_exchange(train)
    if args.mode == "train":
        train = {}
        for backend in args.train_backends.split(","):
            train[backend] = time_train(args.train_config, backend)
        if "hip" in train and "miopen" in train:
            train["speedup_vs_miopen"] = train["hip"]["sampled_proposals_per_s"] / train["miopen"]["sampled_proposals_per_s"]
        train["config"] = "configs/coco_stt.yaml" if args.train_config == "stt" else "configs/coco_lsm.yaml"
        if args.train_config == "stt":
            train["what"] = (f"one STT fine-tune step of the path per iteration (configs/coco_stt.yaml): 3 img/GPU x {args.proposals} proposals "
                             "-> 512 sampled/img, 48-class bank, emb_pred frozen; EmbeddingRes5ROIHeads.forward(targets) (labelling, ROIAlign "
                             "+ Res5 + mean, box predictor, loss_cls + loss_box_reg) + backward (Res5 data + weight gradients, ROIAlign "
                             "backward) + SGD step" + (f"; gradients all-reduced by DDP over {world} ranks" if world > 1 else "")
                             + "; `miopen` = the same module with RES5_BACKEND miopen (torch conv2d autograd)")
        else:
            train["what"] = (f"one LSM training step of the path per iteration: {args.train_images} img/GPU x {args.proposals} proposals -> "
                             f"{args.train_samples} sampled/img; EmbeddingProposalsRes5ROIHeads.forward(targets) (labelling, whole-grid Res5, "
                             "ROIAlign + Res5 + mean, box predictor, losses) + GroundingHead (box branch) + backward (Res5 data + weight "
                             "gradients, ROIAlign backward) + SGD step" + (f"; gradients all-reduced by DDP over {world} ranks" if world > 1 else "")
                             + "; `miopen` = the same module with RES5_BACKEND miopen (torch conv2d autograd)")

    if rank == 0:
        R_local = args.images * args.proposals
        if args.res5 == "hip":
            # the 128x128 f32 NT GEMM (template instance CONV=0): with --res5-dtype fp32 the seven 1x1 convolutions of Res5
            # and the three 121-problem Winograd-domain batched GEMMs; always emb_pred and the similarity GEMM.
            n0, ms0, fl0 = gemm_plain
            n1, ms1, fl1 = gemm_conv
            achieved = fl0 / (ms0 * 1e-3) / 1e12 if ms0 > 0 else None
            survey_res5_flops = 2.0 * 732.2e6 * R_local * args.steps          # 732.2 M MAC per ROI
            roof = {"kernel": "gemm_nt_kernel<float,float,128,128,2,2,2,CONV=0,false,8> (Res5 1x1 convs, "
                              "Winograd-domain batched GEMMs, emb_pred, similarity GEMM)",
                    "bound": "mfma", "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / MFMA_F32_PEAK_TFLOPS if achieved else None,
                    "traffic": recorded_traffic(args, "gemm_nt_kernel<float, float, 128, 128, 2, 2, 2, 0, false, 8"),
                    "traffic_source": traffic_source(args),
                    "traffic_unit": f"HBM-side bytes per launch, averaged over the launches of all instances of this kernel (PMC, profiles/{TRAFFIC_FILE})",
                    "launches_per_step": n0 / args.steps, "avg_launch_ms": ms0 / max(n0, 1),
                    "share_of_step_time": ms0 * 1e-3 / dt2,
                    "executed_flops_per_step": fl0 / args.steps,
                    "survey_flop_count_rate_tflops": survey_res5_flops / ((ms0 + ms1) * 1e-3) / 1e12 if ms0 + ms1 > 0 else None,
                    "note": "survey_flop_count_rate = SURVEY 8d's direct-convolution FLOP count of Res5 over the "
                            "GEMM kernels' time; it exceeds the executed rate because the 3x3 convolutions run in "
                            "the Winograd domain (121 instead of 441 products per tile and channel pair)"}
            if n1:
                roof["direct_conv3x3"] = {"launches_per_step": n1 / args.steps, "avg_launch_ms": ms1 / n1,
                                          "executed_tflops": fl1 / (ms1 * 1e-3) / 1e12}
            if args.res5_dtype == "f16x2":
                # Dominant kernel = the split-operand GEMM (csrc/gemm_split.hip): Res5's 1x1 convolutions and the
                # Winograd-domain batched GEMMs.  It executes three f16 MFMAs per fp32 product block, so it is priced
                # against the dense f16 MFMA peak on the f16 FLOPs it executes (3 x 2MNK); the fp32-equivalent rate
                # (2MNK / time) is reported next to it.
                ns, mss, fls, bys = gemm_split_big
                big_tile = bool(ns)
                if not ns:            # a workload whose launches all stay below the 256x256 threshold: the 128x128 kernel is the dominant one
                    ns, mss, fls, bys = gemm_split
                ach = 3.0 * fls / (mss * 1e-3) / 1e12 if mss > 0 else float("nan")
                na, msa, fla, _ = gemm_split

                def instance(kind):
                    n_, ms_, fl_, by_ = split_kinds[kind]
                    _, key, what = SPLIT_KINDS[kind]
                    if not n_:
                        return None
                    tr = recorded_traffic(args, key)
                    a_ = 3.0 * fl_ / (ms_ * 1e-3) / 1e12 if ms_ > 0 else None
                    return {"what": what, "kernel": key.rstrip("<"), "launches_per_step": n_ / args.steps, "avg_launch_ms": ms_ / n_,
                            "achieved": a_, "frac": a_ / MFMA_16BIT_PEAK_TFLOPS if a_ else None,
                            "algorithmic_bytes": by_ / n_, "traffic": tr,
                            "traffic_over_algorithmic": tr / (by_ / n_) if tr and by_ else None}
                roof = {"kernel": ("gemm_split_big_kernel (csrc/gemm_split_big.hip): the split-operand NT GEMM on its 256x256 tile -- Res5's 1x1 "
                                   "convolutions on pooled rows and the Winograd-domain batched GEMMs; fp32 in / fp32 out, products formed from "
                                   "f16x2 split operands on the f16 matrix pipe") if big_tile else
                                  ("gemm_split_kernel (csrc/gemm_split.hip): the split-operand NT GEMM on its 128x128 tile (this workload has no "
                                   "launch with the 1 024 tiles of 256x256 the big-tile kernel asks for)"),
                        "bound": "mfma", "achieved": ach, "peak": MFMA_16BIT_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / MFMA_16BIT_PEAK_TFLOPS,
                        "traffic": recorded_traffic(args, "gemm_split_big_kernel"),
                        "traffic_source": traffic_source(args),
                        "algorithmic_bytes": bys / max(ns, 1),
                        "traffic_unit": f"HBM-side bytes per launch (memory side of L2: FETCH_SIZE x 2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes; "
                                        f"separate rocprofv3 --pmc passes, profiles/{TRAFFIC_FILE}), averaged over this kernel's launches; "
                                        "`algorithmic_bytes` = every operand read once and every result written once, same average; per launch "
                                        "kind under `instances`",
                        "launches_per_step": ns / args.steps, "avg_launch_ms": mss / max(ns, 1),
                        "share_of_step_time": mss * 1e-3 / dt2,
                        "executed_f16_flops_per_step": 3.0 * fls / args.steps,
                        "fp32_equivalent_tflops": fls / (mss * 1e-3) / 1e12 if mss > 0 else None,
                        "fp32_equivalent_vs_f32_mfma_peak": fls / (mss * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS if mss > 0 else None,
                        "all_split_gemm_launches": {"launches_per_step": na / args.steps, "share_of_step_time": msa * 1e-3 / dt2,
                                                    "achieved": 3.0 * fla / (msa * 1e-3) / 1e12 if msa > 0 else None,
                                                    "what": "the dominant kernel's launches plus the 128x128-tile launches of the same arithmetic "
                                                            "(round 3's lumped figure)"},
                        "f32_mfma_gemms": {"launches_per_step": n0 / args.steps, "share_of_step_time": ms0 * 1e-3 / dt2,
                                           "executed_tflops": achieved},
                        "instances": {k: instance(k) for k in SPLIT_KINDS},
                        "power_cap": {"sustained_mfma_only_tflops": SUSTAINED_F16_MFMA_TFLOPS,
                                      "frac_of_sustained": ach / SUSTAINED_F16_MFMA_TFLOPS,
                                      "what": "a register-only v_mfma_f32_16x16x32_f16 loop with live operands (no LDS, no memory) is held to "
                                              "~1.95-2.0 GHz by the 1 400 W package cap and sustains this rate on MI355X "
                                              "(profiles/r03_mfma_shape_probe.txt); the split GEMM launches run AT the cap "
                                              "(profiles/r03_power_probe_big_tile.txt), so their time is joules / (cap - idle)"},
                        "note": "achieved counts the f16 MFMA FLOPs the kernel executes (three products per fp32 product); "
                                "peak is the dense f16/bf16 MFMA peak"}
            if args.res5_dtype == "bf16":
                # opt-in reduced-precision run: the dominant kernel is the bf16-operand instance of the same
                # template, priced against the dense bf16 MFMA peak; no PMC traffic pass is recorded for it
                nb, msb, flb = gemm_plain_bf16
                nc, msc, flc = gemm_conv_bf16
                ach = flb / (msb * 1e-3) / 1e12 if msb > 0 else float("nan")
                roof = {"kernel": "gemm_nt_kernel<__bf16,float,128,128,...> (Res5 1x1 convs with bf16 operands)",
                        "bound": "mfma", "achieved": ach, "peak": MFMA_16BIT_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / MFMA_16BIT_PEAK_TFLOPS, "traffic": None,
                        "launches_per_step": nb / args.steps, "avg_launch_ms": msb / max(nb, 1),
                        "share_of_step_time": msb * 1e-3 / dt2,
                        "direct_conv3x3_bf16": {"launches_per_step": nc / args.steps, "avg_launch_ms": msc / max(nc, 1),
                                                "executed_tflops": flc / (msc * 1e-3) / 1e12 if msc > 0 else None,
                                                "share_of_step_time": msc * 1e-3 / dt2},
                        "f32_gemms": {"launches_per_step": n0 / args.steps, "share_of_step_time": ms0 * 1e-3 / dt2,
                                      "executed_tflops": achieved}}
        else:
            alg_bytes = args.images * 1024 * 50 * 84 * 4 + R_local * 5 * 4 + R_local * 1024 * 14 * 14 * 4
            achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
            roof = {"kernel": "roi_align_nhwc2nchw_kernel (dominant HAND-WRITTEN kernel; Res5 is MIOpen here)",
                    "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": dom_ms,
                    "algorithmic_bytes_per_launch": alg_bytes}
        out = {
            "metric": "proposals/sec through LSM ROI-head (1333x800, 1000 prop, 1203-class text bank)",
            "value": props_per_step * args.steps / dt2,
            "unit": "proposals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt2 / args.steps * 1e3,
            "per_rank_ms_per_step": per_rank["s2"], "single_steps_ms": step_stats["s2"],
            "timing": {"how": "hipEventElapsedTime between two events on the launch stream around the K timed steps, inside the "
                              "barrier + synchronize bracket, max over ranks (per_rank_ms_per_step lists every rank's); the interpreter's full "
                              "garbage-collection pass is taken in front of the warm-up steps (gc.collect + gc.freeze, un-frozen "
                              "behind the bracket), not at a random step inside it; train.*.ms_per_step_unfrozen_heap is the same step "
                              "with the collector left alone",
                       "wall_ms_per_step": wall["s2"] / args.steps * 1e3},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16 operands / f32 accumulate in Res5 (opt-in reduced precision, not the parity configuration)"
                      if args.res5_dtype == "bf16" else
                      ("f32 (Res5 GEMM products formed from f16x2 split operands on the f16 matrix pipe, f32 accumulate; error "
                       "vs f64 <= 1.5 x the f32-MFMA kernel's + 6e-7 and <= 3e-6 on every tested shape, tests/test_gpu_split_gemm.py)" if args.res5_dtype == "f16x2" and args.res5 == "hip"
                       else "f32") + ("" if args.sim_dtype == "fp32" else " (bf16 similarity operands)")),
            "data": "synthetic",
            "config": {"workload": f"{args.images} img/GPU x {args.proposals} proposals, res4 [B,1024,50,84] fp32, "
                                   f"EmbeddingProposalsRes5ROIHeads._shared_roi_transform (ROIAlign 14x14 -> Res5({args.res5})) -> mean -> "
                                   f"EmbeddingFastRCNNOutputLayers.forward (bbox_pred / emb_pred 2048->{args.dim} -> "
                                   f"similarity GEMM x {args.classes + 1}-row bank, {args.sim_dtype}); forward only",
                       "scope": "S2 (full ROI head incl. Res5), timed through the plugin surface", "images_per_gpu": args.images,
                       "proposals_per_image": args.proposals, "classes": args.classes, "emb_dim": args.dim,
                       "res5_backend": args.res5, "res5_conv3x3": args.conv3x3 if args.res5 == "hip" else "miopen",
                       "res5_block0": args.block0 if args.res5 == "hip" else "miopen",
                       "res5_dtype": args.res5_dtype,
                       "parallelism": f"image-sharded x{world}, no data-path collective"},
            "rccl_ranks": rccl_ranks,
            "process_group": f"{args.dist_backend} x{world}" if dist_on else None,
            "scopes": {"S2_full_head_proposals_per_s": props_per_step * args.steps / dt2,
                       "S1_handwritten_kernels_proposals_per_s": None if args.skip_s1 else props_per_step * args.steps / dt1,
                       "S1_ms_per_step": None if args.skip_s1 else dt1 / args.steps * 1e3, "S1_roi_align_roofline": s1_roi, **variants,
                       "eval_1img": eval_1img},
            "roofline": roof,
        }
        if dt2_f32 is not None:
            out["f32_mfma_reference"] = {"value": props_per_step * args.steps / dt2_f32, "unit": "proposals/s",
                                         "ms_per_step": dt2_f32 / args.steps * 1e3,
                                         "what": "the same job, same run, with RES5_DTYPE fp32 (Res5 GEMMs on v_mfma_f32_32x32x2_f32)"}
        if train is not None:
            out["train"] = train
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
