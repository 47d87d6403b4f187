"""One launch of each GEMM (hand-written and hipBLASLt) at a Res5 shape, for rocprofv3 --pmc runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops
M, N, K = 196000, 512, 2048
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02
for _ in range(5):
    ops.linear(x, w)
    torch.nn.functional.linear(x, w)
torch.cuda.synchronize()
