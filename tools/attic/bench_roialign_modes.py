"""Developer aid: the pooler-contract ROIAlign (NCHW out, 14x14, 1024 channels) in its exact and fast forms at the bench shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from locov_amd import ops
import bench
g = torch.Generator().manual_seed(1992)
B, R = 8, 1000
feat = torch.randn(B, 1024, 50, 84, generator=g).cuda()
rois = torch.cat([torch.cat([torch.full((R, 1), float(i)), bench.synth_boxes(g, R)], 1) for i in range(B)]).cuda()


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


out_gb = B * R * 1024 * 196 * 4 / 1e9
for rep in range(2):
    for mode in ("exact", "fast"):
        ms = t(lambda: ops.roi_align(feat, rois, 14, 1 / 16, 0, True, mode=mode))
        print(f"{mode}: {ms:.3f} ms per {B * R} proposals  ({B * R / ms / 1e3:.2f} M proposals/s, {out_gb / ms:.2f} TB/s of output)", flush=True)
a = ops.roi_align(feat, rois, 14, 1 / 16, 0, True, mode="exact")
b = ops.roi_align(feat, rois, 14, 1 / 16, 0, True, mode="fast")
print("max |fast - exact|", float((a - b).abs().max()))
# by box size: where the time goes
side = ((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2])).sqrt()
for lo, hi in ((0, 112), (112, 224), (224, 448), (448, 2000)):
    sel = rois[(side >= lo) & (side < hi)]
    if len(sel) == 0: continue
    rep = sel.repeat((8000 + len(sel) - 1) // len(sel), 1)[:8000].contiguous()
    print(f"side [{lo},{hi}): {len(sel)} of {len(rois)} rois;  8000 such: exact {t(lambda: ops.roi_align(feat, rep, 14, 1 / 16, 0, True, mode='exact')):.3f} ms"
          f"  fast {t(lambda: ops.roi_align(feat, rep, 14, 1 / 16, 0, True, mode='fast')):.3f} ms", flush=True)
