"""Developer aid (round 5): the even-grid ROIAlign BACKWARD (fp32 atomics into the channels-last map gradient; 800 sampled proposals of ONE size class, 4 images,
1024 channels = the LSM step's launch) per proposal size class -- where its 0.8 ms goes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from locov_amd import ops
g = torch.Generator().manual_seed(7)


def boxes(lo, hi, n=200):
    cx, cy = torch.rand(n, generator=g) * 1333.0, torch.rand(n, generator=g) * 800.0
    side = 2.0 ** (np.log2(lo) + torch.rand(n, generator=g) * (np.log2(hi) - np.log2(lo)))
    aspect = 0.5 + 1.5 * torch.rand(n, generator=g)
    w, h = side * aspect.sqrt(), side / aspect.sqrt()
    return torch.stack([(cx - w / 2).clamp(0, 1333), (cy - h / 2).clamp(0, 800), (cx + w / 2).clamp(0, 1333), (cy + h / 2).clamp(0, 800)], 1).float()


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


out = []
for lo, hi in ((16, 144), (144, 224), (224, 448), (448, 800), (16, 800), (16, 800.001)):
    nimg, per = (3, 512) if hi > 800 else (4, 200)            # (the last entry: the STT step's launch, 3 x 512 proposals)
    rois = torch.cat([torch.cat([torch.full((per, 1), float(i)), boxes(lo, min(hi, 800), per)], 1) for i in range(nimg)]).cuda()
    grad = torch.randn(49 * nimg * per, 1024, generator=g).cuda()
    ms = t(lambda: ops.roi_align_nhwc_bwd(grad, (nimg, 50, 84, 1024), rois, 14, 1.0 / 16, 0, True, bin_stride=2))
    zero = t(lambda: torch.zeros((4, 50, 84, 1024), device="cuda"))
    out.append(f"{lo}-{int(hi)} px{' (STT: 3 x 512)' if hi > 800 else ''}: {ms:.3f} ms")
print("  ".join(out), flush=True)
