"""Developer aid (round 4): the pooler-contract ROIAlign per box-size class (8 000 proposals of ONE class each, 1024 channels) -- where the
LDS-window path (proposals whose pixel rectangle fits the transpose tile) pays; LOCOV_HIP_LIB=tools/liblocov_nowin.so is the direct form alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from locov_amd import ops
g = torch.Generator().manual_seed(7)
feat = torch.randn(8, 1024, 50, 84, generator=g).cuda()


def boxes(lo, hi, n=1000):
    cx, cy = torch.rand(n, generator=g) * 1333.0, torch.rand(n, generator=g) * 800.0
    side = 2.0 ** (np.log2(lo) + torch.rand(n, generator=g) * (np.log2(hi) - np.log2(lo)))
    aspect = 0.5 + 1.5 * torch.rand(n, generator=g)
    w, h = side * aspect.sqrt(), side / aspect.sqrt()
    return torch.stack([(cx - w / 2).clamp(0, 1333), (cy - h / 2).clamp(0, 800), (cx + w / 2).clamp(0, 1333), (cy + h / 2).clamp(0, 800)], 1).float()


def t(fn, n=6, rounds=3):
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


out = [os.environ.get("LOCOV_HIP_LIB", "product")]
for lo, hi in ((16, 112), (112, 224), (224, 448), (448, 800), (16, 800)):
    rois = torch.cat([torch.cat([torch.full((1000, 1), float(i)), boxes(lo, hi)], 1) for i in range(8)]).cuda()
    with torch.no_grad():
        out.append(f"{lo}-{hi} px: {t(lambda: ops.roi_align(feat, rois, 14, 1.0 / 16, 0, True)):.3f} ms")
print("  ".join(out), flush=True)
