"""Developer aid: the wave-specialised split GEMM against the plain one (run twice: LOCOV_SPLIT_WS=0 / 1) and float64."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from locov_amd import ops

out = sys.argv[1]
g = torch.Generator().manual_seed(0)
res = {}
cases = [("k512_res", 70000, 2048, 512, True, True), ("k2048", 70001, 512, 2048, False, True), ("k1024_map", 33600, 2560, 1024, False, False),
         ("k512_small", 40000, 256, 512, True, False)]
for name, M, N, K, has_res, relu in cases:
    x = torch.relu(torch.randn(M, K, generator=g)).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).cuda()
    sc = (torch.rand(N, generator=g) + 0.5).cuda()
    sh = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).cuda() if has_res else None
    wp = ops.split_pack(w)
    y = ops.linear_split(x, wp, sh, scale=sc, residual=r, relu=relu)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        y = ops.linear_split(x, wp, sh, scale=sc, residual=r, relu=relu)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    rows = torch.randint(0, M, (256,), generator=g)
    want = x[rows].double() @ w.double().t() * sc.double() + sh.double()
    if has_res:
        want = want + r[rows].double()
    if relu:
        want = want.clamp_min(0)
    err = float((y[rows].double() - want).abs().max() / want.abs().max())
    print(f"{name}: {dt*1e3:.3f} ms  {2.0*M*N*K/dt/1e12:.0f} TF-eq  err {err:.2e}", flush=True)
    res[name] = y.cpu().numpy()
# batched (Winograd-domain shape)
xb = torch.randn(121, 3000, 512, generator=g).cuda()
wb = ops.split_pack((torch.randn(121, 512, 512, generator=g) * 0.05).cuda())
yb = ops.gemm_nt_batched_split(xb, wb, x_scale=1.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    yb = ops.gemm_nt_batched_split(xb, wb, x_scale=1.0)
torch.cuda.synchronize()
print(f"batched: {(time.perf_counter()-t0)/5*1e3:.3f} ms", flush=True)
res["batched"] = yb[::17].cpu().numpy()
np.savez(out, **res)
