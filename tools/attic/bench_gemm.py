"""Times the hand-written NT GEMM at the Res5 / box-head shapes (developer tool, not the bench contract)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops

def t(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
M = R * 49
for (N, K) in [(512, 1024), (2048, 1024), (2048, 512), (512, 2048), (768, 2048)]:
    m = M if N != 768 else R
    x = torch.randn(m, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02
    ms = t(lambda: ops.linear(x, w))
    ms_t = t(lambda: torch.nn.functional.linear(x, w))
    fl = 2.0 * m * N * K
    print(f"M={m} N={N} K={K}: hip {ms:.3f} ms {fl/ms/1e9:.1f} TF | torch(hipBLASLt) {ms_t:.3f} ms {fl/ms_t/1e9:.1f} TF")
    xb, wb = ops.to_bf16(x), ops.to_bf16(w)
    ms = t(lambda: ops.sim_gemm_bf16(xb, wb))
    print(f"   bf16: hip {ms:.3f} ms {fl/ms/1e9:.1f} TF")
