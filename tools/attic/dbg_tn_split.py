"""Developer timing: the split-arithmetic weight-gradient (TN) GEMM at the training step's shapes (M = 39 200 ROI rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops
M = 39200
for N, K in ((2048, 512), (512, 2048), (2048, 1536), (512, 1024)):
    g = torch.randn(M, N, device="cuda") * 1e-3
    x = torch.randn(M, K, device="cuda").relu_()
    sc = ops.split_scale_from_amax(g)
    ref = (g.double().t() @ x.double())
    out = ops.gemm_tn_split(g, x, None, sc)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    for _ in range(3): ops.gemm_tn_split(g, x, None, sc)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): ops.gemm_tn_split(g, x, None, sc)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"dW [{N},{K}] over {M} rows: {ms:.3f} ms  {6.0 * M * N * K / ms / 1e9:.0f} TF f16  rel err vs fp64 {err:.2e}", flush=True)
