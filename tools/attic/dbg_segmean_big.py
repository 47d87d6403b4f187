import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/locov_amd") else ".")
import torch
from locov_amd import ops
g = torch.Generator().manual_seed(0)
R = 8000; M = 49 * R
xs = ops.split_pack(torch.relu(torch.randn(M, 512, generator=g)).cuda(), 16.0).data
wp = ops.split_pack((torch.randn(2048, 512, generator=g) * 0.05).cuda())
res = ops.split_pack(torch.relu(torch.randn(M, 2048, generator=g)).cuda(), 16.0).data
sh = torch.randn(2048, generator=g).cuda()
def t(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
f = lambda: ops.linear_split_segmean(xs, wp, sh, res, 49, relu=True, residual_roi_major=True, x_is_split=True, x_scale=16.0, residual_is_split=True)
for rep in range(3):
    for big in ("0", "1"):
        os.environ["LOCOV_SPLIT_BIG"] = big
        print("segmean conv3, big =", big, f"{t(f):.3f} ms", flush=True)
