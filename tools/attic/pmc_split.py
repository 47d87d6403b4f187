"""A few launches of the split-operand GEMM at a Res5 shape, for rocprofv3 --pmc runs (tools/attic/pmc_split.sh).
PMC_SPLIT_CASE: conv1 (default) [196000,2048] x [512,2048]^T, A converted in the kernel; conv3_asplit [196000,512] x [2048,512]^T +
residual + ReLU with A pre-split; conv1_asplit the first shape with A pre-split."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops
case = os.environ.get("PMC_SPLIT_CASE", "conv1")
if case.startswith("tn"):
    # weight-gradient (TN) GEMM of the training step: tn_conv3 = dW [2048,512] over 39 200 rows, tn_conv1 = dW [512,2048]
    M = 39200
    N, K = (2048, 512) if case == "tn_conv3" else (512, 2048)
    g = torch.randn(M, N, device="cuda") * 1e-3
    x = torch.randn(M, K, device="cuda").relu_()
    sc = ops.split_scale_from_amax(g)
    for _ in range(5):
        ops.gemm_tn_split(g, x, None, sc)
    torch.cuda.synchronize()
    sys.exit(0)
M = 196000
N, K = (2048, 512) if case == "conv3_asplit" else (512, 2048)
x = torch.randn(M, K, device="cuda").relu_(); w = torch.randn(N, K, device="cuda") * 0.02
ws = ops.split_pack(w)
if case == "conv1":
    f = lambda: ops.linear_split(x, ws)
else:
    xs = ops.split_pack(x, 16.0)
    r = torch.randn(M, N, device="cuda") if case == "conv3_asplit" else None
    f = lambda: ops.linear_split(xs.data, ws, residual=r, relu=True, x_scale=16.0, x_is_split=True)
for _ in range(5):
    f()
torch.cuda.synchronize()
