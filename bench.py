#!/usr/bin/env python
"""bench.py -- proposals/sec through the LSM ROI head on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic input resident in HBM:
`--images` res4 feature maps [B,1024,50,84] (1333x800 images, stride 16) x `--proposals`
boxes each -> ROIAlign 14x14 -> Res5 -> spatial mean -> bbox_pred / emb_pred -> (norm) ->
similarity GEMM against a (`--classes`+1) x `--dim` text bank.  Images shard over
ranks with no data-path collective (inference needs none): weak scaling.

`value` is scope S2 = the full head as the reference executes it (Res5 included).  Scope S1 =
the north-star kernel list only (ROIAlign + mean/FCs/similarity on a stand-in for the Res5
output) is reported beside it in "scopes".  One JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # f32-input MFMA peak
MFMA_BF16_PEAK_TFLOPS = 2500.0 # dense bf16 MFMA peak (only used by the opt-in --res5-dtype bf16 run)
MFMA_BF16_PEAK_TFLOPS = 2500.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
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
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--skip-f32-reference", action="store_true",
                   help="do not also time the f32-MFMA form of the Res5 GEMMs (profile runs)")
    p.add_argument("--skip-s1", action="store_true",
                   help="only the S2 scope (profiling runs: the kernel mix then equals the timed region's)")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline wall time")
    # developer/test knobs: rehearse the multi-process flow on a box with fewer GPUs than ranks
    p.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl")
    p.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (needs --dist-backend gloo)")
    return p.parse_args()


def synth_rois(gen: torch.Generator, n_img: int, r: int, device) -> torch.Tensor:
    """SURVEY.md 8d boxes: centre uniform, log2(side) U[4, log2 800], aspect U[0.5,2], clipped."""
    n = n_img * r
    cx = torch.rand(n, generator=gen) * 1333.0
    cy = torch.rand(n, generator=gen) * 800.0
    side = 2.0 ** (4.0 + torch.rand(n, generator=gen) * (np.log2(800.0) - 4.0))
    aspect = 0.5 + 1.5 * torch.rand(n, generator=gen)
    w, h = side * aspect.sqrt(), side / aspect.sqrt()
    b = torch.stack([(cx - w / 2).clamp(0, 1333), (cy - h / 2).clamp(0, 800),
                     (cx + w / 2).clamp(0, 1333), (cy + h / 2).clamp(0, 800)], dim=1)
    idx = torch.arange(n_img, dtype=torch.float32).repeat_interleave(r)[:, None]
    return torch.cat([idx, b], dim=1).to(torch.float32).to(device)


class Workload:
    def __init__(self, args, device):
        from locov_amd import ops
        from locov_amd.config import get_cfg
        from locov_amd.res5 import build_res5_block
        self.ops, self.args, self.device = ops, args, device
        gen = torch.Generator().manual_seed(1992)          # configs/coco_lsm.yaml:126
        B, R, K, D = args.images, args.proposals, args.classes, args.dim
        self.feat = torch.randn(B, 1024, 50, 84, generator=gen).to(device)
        self.rois = synth_rois(gen, B, R, device)
        res5, _ = build_res5_block(get_cfg())
        self.res5 = res5.to(device).eval()
        self.emb_w = (torch.randn(D, 2048, generator=gen) * 0.01).to(device)
        self.emb_b = torch.zeros(D, device=device)
        self.bbox_w = (torch.randn(4, 2048, generator=gen) * 0.001).to(device)
        self.bbox_b = torch.zeros(4, device=device)
        bank = torch.randn(K + 1, D, generator=gen) * 0.05
        bank[-1] = 0
        self.bank = bank.to(device)
        self.bank16 = ops.to_bf16(self.bank) if args.sim_dtype == "bf16" else None
        self.sim = ops.BF16 if args.sim_dtype == "bf16" else ops.F32
        self.r5_standin = torch.randn(B * R, 2048, 7, 7, generator=gen).clamp_(min=0).to(device)
        self.ev = []        # (start, end) HIP events around the dominant kernel (miopen mode: ROIAlign)
        self.timing = False

    def head(self, x, channels_last=False):
        ops = self.ops
        return ops.box_head(x, self.emb_w, self.emb_b, self.bbox_w, self.bbox_b, self.bank, self.bank16,
                            ops.NORM_NONE, self.sim, channels_last=channels_last)

    @torch.no_grad()
    def step_s2(self, timed=False):
        ops = self.ops
        self.timing = timed
        if self.args.res5 == "hip":
            # channels-last pipeline: even-grid ROIAlign -> Res5 as MFMA GEMMs on pixel rows
            nhwc = ops.nchw_to_nhwc(self.feat)
            # (position-major pixel rows [7,7,R,C]: the 3x3 convs skip their zero-padding taps)
            R = self.rois.shape[0]
            wino = self.args.conv3x3 == "winograd"
            if self.args.res5_dtype == "bf16":
                y = self.res5.forward_from_map(nhwc, self.rois, 14, 1.0 / 16, 0, True, bf16=True)
            elif self.args.block0 == "map" and self.res5.map_path_pays(R, nhwc.shape[0] * 50 * 84):
                # block 0's 1x1 convolutions on the map, ROIAlign pools their outputs (Res5Stage.forward_from_map)
                # (pooled: the stage returns the spatial mean the box head consumes; in split arithmetic it is fused into
                # the last 1x1 convolution)
                y = self.res5.forward_from_map(nhwc, self.rois, 14, 1.0 / 16, 0, True, winograd=wino,
                                               split=self.args.res5_dtype == "f16x2", pooled=True, roi_major=wino)
            else:
                x0 = self.res5.rows_input(49 * R, self.device)
                ops.roi_align_nhwc(nhwc, self.rois, 14, 1.0 / 16, 0, True, bin_stride=2, pos_major=not wino, out=x0)
                y = self.res5.forward_rows(x0, 7, 7, pos_major=not wino, winograd=wino, split=self.args.res5_dtype == "f16x2",
                                           pooled=True)
            out = self.head(y) if y.shape[0] == R else self.head(y.view(7, 7, R, 2048), channels_last=2)
        else:
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            pooled = ops.roi_align(self.feat, self.rois, 14, 1.0 / 16, 0, True)
            if timed:
                e1.record()
                self.ev.append((e0, e1))
            out = self.head(self.res5(pooled))
        self.timing = False
        return out

    @torch.no_grad()
    def step_s1(self):
        self.ops.roi_align(self.feat, self.rois, 14, 1.0 / 16, 0, True)
        return self.head(self.r5_standin)


TRAFFIC_FILE = "r01l_pmc_traffic.json"


def recorded_traffic(args, kernel_key: str):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/<TRAFFIC_FILE>; separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this
    same command, corrected as MI355X_MICROARCH.md prescribes).  Counters cannot be collected from
    inside the timed run, so this is null unless the workload is the one that was profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)) as f:
            rec = json.load(f)
        wl = rec["workload"]
        if any(wl[k] != getattr(args, k) for k in ("images", "proposals", "classes", "dim", "res5", "conv3x3", "block0", "res5_dtype")):
            return None
        for name, v in rec["kernels"].items():
            if kernel_key in name:
                return v["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


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


def cpu_baseline(args, seconds: float):
    """The oracle (a port: the reference's Python cannot run here) timed on this
    host's cores over a bounded sample of the same workload."""
    from oracle import lsm_oracle as oracle
    oracle.build()
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    os.environ["OMP_NUM_THREADS"] = str(ncores)
    rng = np.random.default_rng(1992)
    feat = rng.standard_normal((1, 1024, 50, 84)).astype(np.float32)
    params = oracle.make_res5_params(0)
    head = oracle.synth_head(rng, 2048, args.dim, args.classes)

    def run(n):
        boxes = oracle.synth_boxes(rng, n)
        t0 = time.perf_counter()
        oracle.roi_head_forward(feat, [boxes], params, head)
        return time.perf_counter() - t0

    run(8)                                   # warm-up (thread pools, page-in)
    t_probe = run(50)
    n = int(max(50, min(args.proposals, 50 * seconds / max(t_probe, 1e-3) / 2)))
    times = [run(n) for _ in range(2)]
    t = float(np.median(times))
    return {"value": n / t, "unit": "proposals/s", "cores": ncores, "kind": "port",
            "sample": f"{n} proposals of one 1333x800 image, full head (ROIAlign+Res5+mean+FCs+sim K={args.classes}), "
                      f"median of 2 runs, {t:.2f} s each; torch CPU convs + OpenMP C oracle, "
                      f"cpu={platform.processor() or platform.machine()}"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the hot path has no CPU fallback)")
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        if args.dist_backend == "nccl":       # RCCL over xGMI; only the timing max-reduce and barriers use it
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    from locov_amd import _lib
    _lib.load()
    wl = Workload(args, device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup, on_start=None, **kw):
        for _ in range(warmup):
            fn()
        barrier()
        if on_start is not None:
            on_start()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn(**kw)
        barrier()
        dt = time.perf_counter() - t0
        from locov_amd.sharding import max_over_ranks
        # the slowest rank defines the job's time
        return max_over_ranks(dt, device if args.dist_backend == "nccl" else None)

    props_per_step = args.images * args.proposals * world
    lib = _lib.load()
    # dominant kernel: the library brackets each of its GEMM-kernel launches with HIP events on the
    # launch stream while this is enabled (include/locov_hip.h, locov_gemm_timing_*)
    dt2 = timed(wl.step_s2, args.steps, args.warmup, on_start=lambda: lib.locov_gemm_timing_enable(1), timed=True)

    def gemm_class(cls):
        import ctypes
        n, ms, fl = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        _lib.check(lib.locov_gemm_timing_read(cls, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)))
        return n.value, ms.value, fl.value
    gemm_plain, gemm_conv = gemm_class(0), gemm_class(1)
    gemm_plain_bf16, gemm_conv_bf16 = gemm_class(3), gemm_class(4)
    gemm_split = gemm_class(5)
    lib.locov_gemm_timing_enable(0)
    # the same job with the Res5 GEMMs on the f32 MFMA (reported beside the headline when that is the split path)
    dt2_f32 = None
    if args.res5 == "hip" and args.res5_dtype == "f16x2" and not args.skip_f32_reference:
        args.res5_dtype = "fp32"
        dt2_f32 = timed(wl.step_s2, args.steps, args.warmup, timed=True)
        args.res5_dtype = "f16x2"
    dom_ms = float(np.mean([a.elapsed_time(b) for a, b in wl.ev])) if wl.ev else float("nan")
    dt1 = timed(wl.step_s1, args.steps, args.warmup) if not args.skip_s1 else float("nan")

    if rank == 0:
        R_local = args.images * args.proposals
        if args.res5 == "hip":
            # Dominant kernel = the 128x128 NT GEMM (template instance CONV=0): the seven 1x1 convolutions of
            # Res5, the three 121-problem Winograd-domain batched GEMMs, emb_pred and the similarity GEMM.
            # achieved = the FLOPs those launches execute (2*M*N*K each, = SURVEY 8d's count for the 1x1
            # convs / FCs; the Winograd GEMMs execute 121/441 of SURVEY's 3x3 count) / their summed
            # durations, both taken per launch inside the timed region.
            n0, ms0, fl0 = gemm_plain
            n1, ms1, fl1 = gemm_conv
            achieved = fl0 / (ms0 * 1e-3) / 1e12 if ms0 > 0 else float("nan")
            survey_res5_flops = 2.0 * 732.2e6 * R_local * args.steps          # 732.2 M MAC per ROI
            roof = {"kernel": "gemm_nt_kernel<float,float,128,128,2,2,2,CONV=0,false,8> (Res5 1x1 convs, "
                              "Winograd-domain batched GEMMs, emb_pred, similarity GEMM)",
                    "bound": "mfma", "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / MFMA_F32_PEAK_TFLOPS,
                    "traffic": recorded_traffic(args, "gemm_nt_kernel<float, float, 128, 128, 2, 2, 2, 0, false, 8>"),
                    "traffic_unit": f"HBM-side bytes per launch, averaged over this kernel's launches (PMC, profiles/{TRAFFIC_FILE})",
                    "launches_per_step": n0 / args.steps, "avg_launch_ms": ms0 / max(n0, 1),
                    "share_of_step_time": ms0 * 1e-3 / dt2,
                    "executed_flops_per_step": fl0 / args.steps,
                    "survey_flop_count_rate_tflops": survey_res5_flops / ((ms0 + ms1) * 1e-3) / 1e12,
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
                ns, mss, fls = gemm_split
                ach = 3.0 * fls / (mss * 1e-3) / 1e12 if mss > 0 else float("nan")
                roof = {"kernel": "gemm_split_kernel (Res5 1x1 convs, Winograd-domain batched GEMMs; fp32 in/out, f16x2 split "
                                  "operands on the f16 matrix pipe)",
                        "bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / MFMA_BF16_PEAK_TFLOPS,
                        "traffic": recorded_traffic(args, "gemm_split_kernel"),
                        "traffic_unit": f"HBM-side bytes per launch, averaged over this kernel's launches (PMC, profiles/{TRAFFIC_FILE})",
                        "launches_per_step": ns / args.steps, "avg_launch_ms": mss / max(ns, 1),
                        "share_of_step_time": mss * 1e-3 / dt2,
                        "executed_f16_flops_per_step": 3.0 * fls / args.steps,
                        "fp32_equivalent_tflops": fls / (mss * 1e-3) / 1e12 if mss > 0 else None,
                        "fp32_equivalent_vs_f32_mfma_peak": fls / (mss * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS if mss > 0 else None,
                        "f32_mfma_gemms": {"launches_per_step": n0 / args.steps, "share_of_step_time": ms0 * 1e-3 / dt2,
                                           "executed_tflops": achieved},
                        "note": "achieved counts the f16 MFMA FLOPs the kernel executes (three products per fp32 product); "
                                "peak is the dense f16/bf16 MFMA peak"}
            if args.res5_dtype == "bf16":
                # opt-in reduced-precision run: the dominant kernel is the bf16-operand instance of the same
                # template, priced against the dense bf16 MFMA peak; no PMC traffic pass is recorded for it
                nb, msb, flb = gemm_plain_bf16
                nc, msc, flc = gemm_conv_bf16
                ach = flb / (msb * 1e-3) / 1e12 if msb > 0 else float("nan")
                roof = {"kernel": "gemm_nt_kernel<__bf16,float,128,128,...> (Res5 1x1 convs with bf16 operands)",
                        "bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / MFMA_BF16_PEAK_TFLOPS, "traffic": None,
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
            roof = {"kernel": "roi_align_nchw_kernel<fwd> (dominant HAND-WRITTEN kernel; Res5 is MIOpen here)",
                    "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": dom_ms,
                    "algorithmic_bytes_per_launch": alg_bytes}
        out = {
            "metric": "proposals/sec through LSM ROI-head (1333x800, 1000 prop, 1203-class text bank)",
            "value": props_per_step * args.steps / dt2,
            "unit": "proposals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt2 / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16 operands / f32 accumulate in Res5 (opt-in reduced precision, not the parity configuration)"
                      if args.res5_dtype == "bf16" else
                      ("f32 (Res5 GEMM products formed from f16x2 split operands on the f16 matrix pipe, f32 accumulate; error "
                       "vs f64 <= the f32-MFMA path's, tests/test_gpu_split_gemm.py)" if args.res5_dtype == "f16x2" and args.res5 == "hip"
                       else "f32") + ("" if args.sim_dtype == "fp32" else " (bf16 similarity operands)")),
            "data": "synthetic",
            "config": {"workload": f"{args.images} img/GPU x {args.proposals} proposals, res4 [B,1024,50,84] fp32, "
                                   f"ROIAlign 14x14 -> Res5({args.res5}) -> mean -> bbox_pred/emb_pred(2048->{args.dim}) -> "
                                   f"similarity GEMM x {args.classes + 1}-row bank ({args.sim_dtype}); forward only",
                       "scope": "S2 (full ROI head incl. Res5)", "images_per_gpu": args.images,
                       "proposals_per_image": args.proposals, "classes": args.classes, "emb_dim": args.dim,
                       "res5_backend": args.res5, "res5_conv3x3": args.conv3x3 if args.res5 == "hip" else "miopen",
                       "res5_block0": args.block0 if args.res5 == "hip" else "miopen",
                       "res5_dtype": args.res5_dtype,
                       "parallelism": f"image-sharded x{world}, no collective"},
            "scopes": {"S2_full_head_proposals_per_s": props_per_step * args.steps / dt2,
                       "S1_handwritten_kernels_proposals_per_s": None if args.skip_s1 else props_per_step * args.steps / dt1,
                       "S1_ms_per_step": None if args.skip_s1 else dt1 / args.steps * 1e3},
            "roofline": roof,
        }
        if dt2_f32 is not None:
            out["f32_mfma_reference"] = {"value": props_per_step * args.steps / dt2_f32, "unit": "proposals/s",
                                         "ms_per_step": dt2_f32 / args.steps * 1e3,
                                         "what": "the same job, same run, with --res5-dtype fp32 (Res5 GEMMs on v_mfma_f32_32x32x2_f32)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
