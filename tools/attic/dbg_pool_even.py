"""Developer timing: the even-grid channels-last ROIAlign (block 0's pooler on the map GEMM's output) at 512 and 2048 channels, 8 000 proposals."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops
g = torch.Generator().manual_seed(0)
Nimg, H, W, R = 8, 50, 84, 8000
fmap = torch.randn(Nimg, H, W, 2560, generator=g).cuda()
wh = torch.rand(R, 2, generator=g) ** 2 * torch.tensor([W * 16.0, H * 16.0]) * 0.8 + 16.0
xy = torch.rand(R, 2, generator=g) * torch.tensor([W * 16.0, H * 16.0]) * 0.6
rois = torch.cat([(torch.arange(R) // 1000).float()[:, None], xy, xy + wh], dim=1).cuda()


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for lo, hi in ((0, 512), (512, 2560)):
    f = lambda: ops.roi_align_nhwc(fmap[..., lo:hi], rois, 14, 1.0 / 16, 0, True, bin_stride=2)
    print(f"channels {hi - lo}: {t(f):.3f} ms", flush=True)
