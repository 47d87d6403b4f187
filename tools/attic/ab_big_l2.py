"""Developer aid (round 4): the 256x256 split GEMM's launch kinds at the bench's sizes, one library per process (LOCOV_HIP_LIB selects a
tools/liblocov_<tag>.so variant: tile order / cache policy of the operand DMAs).  Prints ms per launch, best of 3 rounds of 6."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops
g = torch.Generator().manual_seed(0)
R = int(os.environ.get("R", 8000))
M = 49 * R


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


y2 = ops.split_pack(torch.relu(torch.randn(M, 512, generator=g)).cuda(), 16.0).data
w3 = ops.split_pack((torch.randn(2048, 512, generator=g) * 0.05).cuda())
res_s = ops.split_pack(torch.relu(torch.randn(M, 2048, generator=g)).cuda(), 16.0).data
w1 = ops.split_pack((torch.randn(512, 2048, generator=g) * 0.02).cuda())
out = {}
out["conv3 (split res -> split out)"] = t(lambda: ops.linear_split(y2, w3, residual=res_s, relu=True, x_is_split=True, residual_is_split=True, out_split=True))
out["conv3 mean-fused"] = t(lambda: ops.linear_split_segmean(y2, w3, None, res_s, 49, relu=True, x_is_split=True, residual_is_split=True, residual_roi_major=True))
out["conv1 pre-split (plain)"] = t(lambda: ops.linear_split(res_s, w1, relu=True, x_is_split=True))
del y2, res_s
x = torch.relu(torch.randn(M, 512, generator=g)).cuda()
u = ops.split_pack(ops.winograd_pack_weight((torch.randn(512, 512, 3, 3, generator=g) * 0.02).cuda()))
out["winograd conv3x3 (in transform + 121 batched GEMMs + out transform)"] = t(lambda: ops.winograd_conv3x3(x, u, relu=True, roi_major=True, in_roi_major=True))
print(os.environ.get("LOCOV_HIP_LIB", "product"), " ".join(f"{k}: {v:.3f} ms;" for k, v in out.items()), flush=True)
