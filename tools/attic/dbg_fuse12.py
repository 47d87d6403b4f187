"""Developer timing: a bottleneck's conv1 + conv2 as two calls (linear_split -> winograd_conv3x3) against the one-call form whose
conv1 epilogue applies the Winograd input transform (ops.conv1x1_winograd_conv3x3), at the Res5 shape, 8 000 proposals."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
g = torch.Generator().manual_seed(0)
xs = ops.split_pack(torch.relu(torch.randn(49 * R, 2048, generator=g)).cuda(), 16.0).data
w1 = ops.split_pack((torch.randn(512, 2048, generator=g) * 0.02).cuda())
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


def two():
    y = ops.linear_split(xs, w1, b1, scale=s1, relu=True, x_is_split=True, x_scale=16.0)
    return ops.winograd_conv3x3(y, u, scale=s2, shift=b2, relu=True, roi_major=True, in_roi_major=True, out_split_scale=16.0)


def one():
    return ops.conv1x1_winograd_conv3x3(xs, w1, b1, u, scale1=s1, scale2=s2, shift2=b2, relu=True, x_scale=16.0, out_split_scale=16.0)


print("equal:", torch.equal(two(), one()))
for rep in range(3):
    print(f"two calls {t(two):.3f} ms   one call {t(one):.3f} ms   conv1 alone {t(lambda: ops.linear_split(xs, w1, b1, scale=s1, relu=True, x_is_split=True, x_scale=16.0)):.3f} ms", flush=True)
