"""Developer aid (round 4): the even-grid channels-last ROIAlign of block 0's shortcut (2048 of the map GEMM's 2560 output channels,
8 x 1000 bench proposals) -- ms per launch; LOCOV_HIP_LIB / LOCOV_ROIALIGN_SLICES select the variant (tools/attic/ab_pool.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from locov_amd import ops
g = torch.Generator().manual_seed(1992)
Nimg, H, W, R = 8, 50, 84, 1000
fmap = torch.randn(Nimg, H, W, 2560, generator=g).cuda()
rois = torch.cat([torch.cat([torch.full((R, 1), float(i)), bench.synth_boxes(g, R)], 1) for i in range(Nimg)]).cuda()
out = torch.empty(49 * Nimg * R, 2048, device="cuda")


def t(fn, n=10, rounds=3):
    for _ in range(3): fn()
    best = 1e9
    for _ in range(rounds):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best


f = lambda: ops.roi_align_nhwc(fmap[..., 512:], rois, 14, 1.0 / 16, 0, True, bin_stride=2, out=out)
print(os.environ.get("LOCOV_HIP_LIB", "product"), "slices", os.environ.get("LOCOV_ROIALIGN_SLICES", "default"), f"2048-channel pooler: {t(f):.3f} ms", flush=True)
