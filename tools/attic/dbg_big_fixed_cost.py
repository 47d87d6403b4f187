"""Developer timing: per-tile fixed cost (workgroup turnover + prologue + epilogue) of the 256x256 split GEMM: time over K at fixed M, N."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops
M = 392000
os.environ["LOCOV_SPLIT_BIG"] = "1"
for N in (512, 2048):
    rows = []
    for K in (64, 128, 256, 512, 1024, 2048):
        g = torch.Generator().manual_seed(K)
        xs = ops.split_pack(torch.relu(torch.randn(M, K, generator=g)).cuda(), 16.0).data
        wp = ops.split_pack((torch.randn(N, K, generator=g) * 0.05).cuda())
        f = lambda: ops.linear_split(xs, wp, relu=True, x_scale=16.0, x_is_split=True)
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): f()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        tiles_per_cu = (M + 255) // 256 * (N // 256) / 256
        rows.append((K, ms))
        print(f"N={N} K={K:5d}: {ms:.3f} ms   {ms * 1e3 / tiles_per_cu:.1f} us per tile and CU   {6.0 * M * N * K / ms / 1e9:.0f} TF", flush=True)
        del xs
    (k0, t0), (k1, t1) = rows[0], rows[-1]
    slope = (t1 - t0) / (k1 - k0)
    print(f"N={N}: slope {slope * 1e3:.3f} us of launch time per unit of K, intercept {t0 - slope * k0:.3f} ms")
