"""Developer aid: N training steps of the path and nothing else (for rocprofv3 runs; see tools/profile_train.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
import bench

args = bench.parse(sys.argv[1:])
device = torch.device("cuda", 0)
tw = bench.TrainWorkload(args, device, "hip", 1)
for _ in range(args.warmup):
    tw.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    tw.step()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / args.steps * 1e3:.2f} ms per training step")
