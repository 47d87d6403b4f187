"""Developer experiment: even-grid ROIAlign (map path shapes) with proposals in random vs spatially sorted order."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from locov_amd import ops
gen = torch.Generator().manual_seed(1992)
B, R = 8, 1000
dev = torch.device("cuda")
g = torch.randn(B, 50, 84, 2560, generator=gen).to(dev)
rois = bench.synth_rois(gen, B, R, dev)
cx, cy = (rois[:, 1] + rois[:, 3]) * 0.5, (rois[:, 2] + rois[:, 4]) * 0.5
def morton(ix, iy):
    k = torch.zeros_like(ix)
    for b in range(6):
        k |= ((ix >> b) & 1) << (2 * b) | ((iy >> b) & 1) << (2 * b + 1)
    return k
key_rowmajor = rois[:, 0] * 1e6 + (cy / 64).floor() * 1e3 + cx / 16
key_morton = rois[:, 0].long() * 4096 + morton((cx / 32).long(), (cy / 32).long())
orders = {"as given": torch.arange(B * R, device=dev), "row-major tiles": key_rowmajor.argsort(), "morton": key_morton.argsort()}
def t(f, n=10):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, o in orders.items():
    r = rois[o].contiguous()
    a = t(lambda: ops.roi_align_nhwc(g[..., :512], r, 14, 1 / 16, 0, True, bin_stride=2, pos_major=True))
    b = t(lambda: ops.roi_align_nhwc(g[..., 512:], r, 14, 1 / 16, 0, True, bin_stride=2, pos_major=True))
    print(f"{name:16s} 512 ch {a:.3f} ms   2048 ch {b:.3f} ms")
