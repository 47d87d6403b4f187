"""developer A/B (round 5): variants of the LSM / STT training step INTERLEAVED in one process on one box -- blocks of steps of each
variant in turn, several rounds -- so that clock / thermal drift and box-to-box spread (+-2 %) cancel; per-step CUDA-event times,
median per variant.  Variants are run-time switches: the training range guard (sync | deferred) and the one-launch operand
preparation (LOCOV_RES5_PREP).
usage: python tools/ab_train_inproc.py [--train-config lsm|stt] [--steps 20 (per block)] [--rounds 6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gc
import numpy as np
import torch
import bench
from locov_amd import res5 as res5_mod

rounds = 6
argv = sys.argv[1:]
if "--rounds" in argv:
    i = argv.index("--rounds")
    rounds = int(argv[i + 1])
    del argv[i:i + 2]
args = bench.parse(argv)
device = torch.device("cuda", 0)
tw = bench.TrainWorkload(args, device, "hip", 1)
heads = tw.heads
variants = {"sync,prep": ("sync", True), "deferred,prep": ("deferred", True), "sync,noprep": ("sync", False), "deferred,noprep": ("deferred", False)}


def use(name):
    guard, prep = variants[name]
    heads.res5_train_guard = guard
    res5_mod._ONE_LAUNCH_PREP = prep


times = {k: [] for k in variants}
for name in variants:
    use(name)
    for _ in range(8):
        tw.step()
gc.collect()
gc.freeze()
for r in range(rounds):
    for name in variants:
        use(name)
        for _ in range(3):
            tw.step()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        evs[0].record()
        for i in range(args.steps):
            tw.step()
            evs[i + 1].record()
        torch.cuda.synchronize()
        times[name] += [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
for name, t in times.items():
    t = np.array(t)
    print(f"{name:18s} median {np.median(t):6.2f} ms  mean {t.mean():6.2f}  p10 {np.percentile(t, 10):6.2f}  p90 {np.percentile(t, 90):6.2f}  ({len(t)} steps)")
