"""Developer aid: split GEMM with A converted in the kernel vs A pre-split (staged by LDS DMA)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops
g = torch.Generator().manual_seed(0)
for name, M, N, K, has_res in (("conv3 K=512 N=2048 +res", 196000, 2048, 512, True), ("wino-like K=512 N=512", 968000, 512, 512, False), ("conv1 K=2048 N=512", 392000, 512, 2048, False),
                               ("small K=512", 40000, 512, 512, False)):
    x = torch.relu(torch.randn(M, K, generator=g)).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).cuda()
    r = torch.randn(M, N, generator=g).cuda() if has_res else None
    wp = ops.split_pack(w)
    xs = ops.split_pack(x, 16.0)
    def run(f, n=5):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): y = f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, y
    t0, y0 = run(lambda: ops.linear_split(x, wp, residual=r, relu=True))
    t1, y1 = run(lambda: ops.linear_split(xs.data, wp, residual=r, relu=True, x_scale=16.0, x_is_split=True))
    if os.environ.get("SAVE"):
        import numpy as np
        np.save(os.environ["SAVE"] + name.split()[0] + str(K) + ".npy", y1[::97].cpu().numpy())
    print(f"{name}: in-kernel split {t0*1e3:.3f} ms ({2.0*M*N*K/t0/1e12:.0f} TF-eq)   pre-split A {t1*1e3:.3f} ms ({2.0*M*N*K/t1/1e12:.0f} TF-eq)   equal {torch.equal(y0, y1)}  maxdiff {float((y0-y1).abs().max()):.2e}", flush=True)
