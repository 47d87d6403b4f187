"""Developer timing of the bf16 NT GEMM (fp32 accumulate / output) at Res5-size shapes, vs torch (hipBLASLt)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from locov_amd import ops

def t(f, n=10):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (M, N, K) in [(196000, 512, 2048), (196000, 2048, 512), (196000, 2048, 1024), (196000, 512, 512)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02
    xb, wb = ops.to_bf16(x), ops.to_bf16(w)
    xt, wt = x.to(torch.bfloat16), w.to(torch.bfloat16)
    fl = 2.0 * M * N * K
    a = t(lambda: ops.sim_gemm_bf16(xb, wb)); b = t(lambda: torch.nn.functional.linear(xt, wt))
    print(f"M={M} N={N} K={K}: locov bf16 {a:.3f} ms / {fl/a/1e9:.0f} TF   hipblaslt bf16 (bf16 out) {b:.3f} ms / {fl/b/1e9:.0f} TF")
