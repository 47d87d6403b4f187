"""Developer aid: cost of the split-layout epilogue (LOCOV_EPI_OUT_SPLIT / RES_SPLIT) on conv3's shape, and the conv1 it speeds up."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops
g = torch.Generator().manual_seed(0)
R = 8000
M = 49 * R


def t(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


y2 = ops.split_pack(torch.relu(torch.randn(M, 512, generator=g)).cuda(), 16.0).data
w3 = ops.split_pack((torch.randn(2048, 512, generator=g) * 0.05).cuda())
res = torch.relu(torch.randn(M, 2048, generator=g)).cuda()
res_s = ops.split_pack(res, 16.0).data
w1 = ops.split_pack((torch.randn(512, 2048, generator=g) * 0.02).cuda())
for rep in range(2):
    print(f"conv3 fp32 res -> fp32 out   {t(lambda: ops.linear_split(y2, w3, residual=res, relu=True, x_is_split=True)):.3f} ms")
    print(f"conv3 fp32 res -> split out  {t(lambda: ops.linear_split(y2, w3, residual=res, relu=True, x_is_split=True, out_split=True)):.3f} ms")
    print(f"conv3 split res -> fp32 out  {t(lambda: ops.linear_split(y2, w3, residual=res_s, relu=True, x_is_split=True, residual_is_split=True)):.3f} ms")
    print(f"conv3 split res -> split out {t(lambda: ops.linear_split(y2, w3, residual=res_s, relu=True, x_is_split=True, residual_is_split=True, out_split=True)):.3f} ms")
    print(f"conv1 converting             {t(lambda: ops.linear_split(res, w1, relu=True)):.3f} ms")
    print(f"conv1 pre-split              {t(lambda: ops.linear_split(res_s, w1, relu=True, x_is_split=True)):.3f} ms", flush=True)
