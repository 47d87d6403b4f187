"""Developer timing: block 0's pooler + conv2 as two calls (roi_align_nhwc -> winograd_conv3x3) against ops.roi_align_winograd_conv3x3
(the ROIAlign workgroup writes the Winograd input transform itself), 8 images x 1000 proposals, 512 of 2560 map channels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops
g = torch.Generator().manual_seed(0)
Nimg, H, W, R = 8, 50, 84, 8000
fmap = torch.randn(Nimg, H, W, 2560, generator=g).cuda()
feat = fmap[..., :512]
wh = torch.rand(R, 2, generator=g) ** 2 * torch.tensor([W * 16.0, H * 16.0]) * 0.8 + 16.0
xy = torch.rand(R, 2, generator=g) * torch.tensor([W * 16.0, H * 16.0]) * 0.6
rois = torch.cat([(torch.arange(R) // 1000).float()[:, None], xy, xy + wh], dim=1).cuda()
s1, b1 = (torch.rand(512, generator=g) + 0.5).cuda(), (torch.randn(512, generator=g) * 0.3).cuda()
u = ops.split_pack(ops.winograd_pack_weight((torch.randn(512, 512, 3, 3, generator=g) * 0.02).cuda()))
s2, b2 = (torch.rand(512, generator=g) + 0.5).cuda(), (torch.randn(512, generator=g) * 0.1).cuda()


def t(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def pool():
    return ops.roi_align_nhwc(feat, rois, 14, 1.0 / 16, 0, True, bin_stride=2, ch_scale=s1, ch_shift=b1, relu=True).view(49 * R, 512)


def two():
    return ops.winograd_conv3x3(pool(), u, scale=s2, shift=b2, relu=True, roi_major=True, in_roi_major=True, out_split_scale=16.0)


def one():
    return ops.roi_align_winograd_conv3x3(feat, rois, 14, 1.0 / 16, 0, True, u, ch_scale=s1, ch_shift=b1, scale2=s2, shift2=b2, relu=True,
                                          out_split_scale=16.0)


print("equal:", torch.equal(two(), one()))
for rep in range(3):
    print(f"two calls {t(two):.3f} ms   one call {t(one):.3f} ms   pooler alone {t(pool):.3f} ms", flush=True)
