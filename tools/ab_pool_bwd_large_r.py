"""Developer aid: the even-grid ROIAlign backward by tile ownership against the scatter form as the number of proposals
grows (the ownership form's workgroups scan ALL proposals for the ones that reach their tile).  usage: python3 tools/ab_pool_bwd_large_r.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from locov_amd import ops

dev = torch.device("cuda:0")
N, H, W, C = 4, 50, 84, 1024
gen = torch.Generator().manual_seed(1)
for per in (200, 1000, 3000):
    rois = torch.cat([torch.cat([torch.full((per, 1), float(i)), bench.synth_boxes(gen, per)], 1) for i in range(N)]).to(dev)
    G = torch.randn(49 * N * per, C, device=dev)
    out = {}
    for tiles in ("1", "0"):
        os.environ["LOCOV_POOL_BWD_TILES"] = tiles
        for _ in range(3):
            ops.roi_align_nhwc_bwd(G, (N, H, W, C), rois, 14, 1 / 16, 0, True, bin_stride=2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            ops.roi_align_nhwc_bwd(G, (N, H, W, C), rois, 14, 1 / 16, 0, True, bin_stride=2)
        e1.record()
        torch.cuda.synchronize()
        out[tiles] = e0.elapsed_time(e1) / 10
    print(f"{N} images x {per} proposals ({N * per} in all), {C} channels: ownership {out['1']:.3f} ms, scatter {out['0']:.3f} ms")
