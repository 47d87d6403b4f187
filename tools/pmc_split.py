"""A few launches of the split-operand GEMM at a Res5 shape, for rocprofv3 --pmc runs (tools/pmc_split.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locov_amd import ops
M, N, K = 196000, 512, 2048
x = torch.randn(M, K, device="cuda").relu_(); w = torch.randn(N, K, device="cuda") * 0.02
ws = ops.split_pack(w)
for _ in range(5):
    ops.linear_split(x, ws)
torch.cuda.synchronize()
