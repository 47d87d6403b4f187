"""developer diagnostic (round 5): where does the GPU wait for the HOST inside an un-profiled LSM training step?  CUDA events
are recorded at the seams of the step (they cost ~1 us each and no synchronisation): the time between "the grid call's last
kernel finished" and "the proposals' first kernel started" is GPU idle time whenever the host was late, etc.
usage: python tools/train_gaps.py [--steps 30]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np
import torch
import bench
from locov_amd import res5_train
from locov_amd.roi_heads import roi_emb_heads as H

args = bench.parse(sys.argv[1:])
device = torch.device("cuda", 0)
tw = bench.TrainWorkload(args, device, "hip", 1)
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e, time.perf_counter()))


def wrap(mod, fname, before, after):
    f = getattr(mod, fname)
    def g(*a, **k):
        if before: mark(before)
        out = f(*a, **k)
        if after: mark(after)
        return out
    setattr(mod, fname, g)


wrap(res5_train, "grid_segment", "grid_fwd_begin", "grid_fwd_enqueued")
wrap(res5_train, "roi_segment", "roi_fwd_begin", "roi_fwd_enqueued")
lf = H.SampleAllROIHeads._label_finish
def label_finish(self, st):
    mark("label_wait_begin")
    out = lf(self, st)
    mark("label_finish_done")
    return out
H.SampleAllROIHeads._label_finish = label_finish
lb = H.SampleAllROIHeads._label_begin
def label_begin(self, *a):
    mark("label_begin")
    out = lb(self, *a)
    mark("label_enqueued")
    return out
H.SampleAllROIHeads._label_begin = label_begin
rg = H.ops.RangeGuard.raised
def raised(self):
    mark("guard_wait_begin")
    out = rg(self)
    mark("guard_wait_done")
    return out
H.ops.RangeGuard.raised = raised
fb = tw.forward_backward
def forward_backward(*a, **k):
    mark("fwd_begin")
    out = fb(*a, **k)
    mark("bwd_enqueued")
    return out
tw.forward_backward = forward_backward

for _ in range(8):
    tw.step()
torch.cuda.synchronize()
rows = []
for _ in range(args.steps):
    marks.clear()
    mark("step_begin")
    tw.step()
    mark("step_enqueued")
    torch.cuda.synchronize()
    t0e, t0h = marks[0][1], marks[0][2]
    rows.append([(n, t0e.elapsed_time(e), (th - t0h) * 1e3) for n, e, th in marks])
names = [r[0] for r in rows[0]]
gpu = np.median(np.array([[x[1] for x in r] for r in rows]), axis=0)
host = np.median(np.array([[x[2] for x in r] for r in rows]), axis=0)
print(f"{'mark':24s} {'GPU reaches it (ms)':>20s} {'host passes it (ms)':>20s}   (medians of {len(rows)} steps; an event's GPU time = when everything enqueued before it has finished)")
for n, g, h in zip(names, gpu, host):
    print(f"{n:24s} {g:20.3f} {h:20.3f}")
