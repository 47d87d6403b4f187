"""Developer aid: N training steps of the path and nothing else (for rocprofv3 runs; see tools/profile_train.sh).  Prints the mean over
the bracket and, from one CUDA event per step, the median / 10th / 90th percentile of the single steps (the mean of a short bracket is
moved by one garbage collection or one clock ramp; the median is what a kernel change shows up in)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gc
import time
import numpy as np
import torch
import bench

args = bench.parse(sys.argv[1:])
device = torch.device("cuda", 0)
tw = bench.TrainWorkload(args, device, "hip", 1)
for _ in range(args.warmup):
    tw.step()
gc.collect()
gc.freeze()
torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
t0 = time.perf_counter()
evs[0].record()
for i in range(args.steps):
    tw.step()
    evs[i + 1].record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / args.steps * 1e3
per = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)])
print(f"{wall:.2f} ms per training step (median {np.median(per):.2f}, p10 {np.percentile(per, 10):.2f}, p90 {np.percentile(per, 90):.2f}, {args.steps} steps)")
