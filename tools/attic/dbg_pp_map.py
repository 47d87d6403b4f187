"""Developer aid: element placement of gemm_split_pp_kernel's epilogue (A = 0, so the output must equal relu-less residual)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops
M, N, K = 196000, 2048, 512
x = torch.zeros(M, K).cuda()
w = torch.randn(N, K).cuda() * 0.05
wp = ops.split_pack(w)
xs = ops.split_pack(x, 16.0)
r = (torch.arange(M, dtype=torch.float32)[:, None] % 2048 * 2048 + torch.arange(N, dtype=torch.float32)[None, :]).cuda()
y = ops.linear_split(xs.data, wp, residual=r, relu=False, x_scale=16.0, x_is_split=True)
bad = (y != r)
print("mismatches", int(bad.sum()), "of", y.numel())
if bad.any():
    idx = bad.nonzero()[:40]
    for m, n in idx.tolist():
        v = float(y[m, n]); print(f"out[{m},{n}] = residual[{int(v)//2048},{int(v)%2048}]")
