"""Developer aid (round 4): the pooler-contract ROIAlign (NCHW in, [R,1024,14,14] out) on the bench's 8 x 1000 proposals -- ms per call
(incl. the 0.07 ms channels-last copy of the map); LOCOV_HIP_LIB selects a store-policy variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from locov_amd import ops
g = torch.Generator().manual_seed(1992)
feat = torch.randn(8, 1024, 50, 84, generator=g).cuda()
rois = torch.cat([torch.cat([torch.full((1000, 1), float(i)), bench.synth_boxes(g, 1000)], 1) for i in range(8)]).cuda()


def t(fn, n=8, rounds=3):
    for _ in range(2): fn()
    best = 1e9
    for _ in range(rounds):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best


with torch.no_grad():
    print(os.environ.get("LOCOV_HIP_LIB", "product"), f"contract ROIAlign: {t(lambda: ops.roi_align(feat, rois, 14, 1.0 / 16, 0, True)):.3f} ms", flush=True)
