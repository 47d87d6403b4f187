"""Developer aid: 128x128 (two workgroups per CU) vs 256x128 (one 8-wave workgroup per CU) tile of the split GEMM, same process,
alternating arms.  LOCOV_SPLIT_BIG is read by the launcher at every call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locov_amd import ops
g = torch.Generator().manual_seed(0)
R = int(os.environ.get("R", 8000))


def arm(big, f, n=6):
    os.environ["LOCOV_SPLIT_BIG"] = str(big)
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): y = f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n, y


for name, M, N, K, has_res, pre in (("conv3 K=512 N=2048 +res presplit", 49 * R, 2048, 512, True, True),
                                    ("conv1 K=2048 N=512 presplit", 49 * R, 512, 2048, False, True),
                                    ("conv1 K=2048 N=512 converting", 49 * R, 512, 2048, False, False),
                                    ("map K=1024 N=2560 converting", 4200 * (R // 1000), 2560, 1024, False, False),
                                    ("ragged M K=512 N=512 presplit", 49 * 777 + 13, 512, 512, True, True)):
    x = torch.relu(torch.randn(M, K, generator=g)).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).cuda()
    r = torch.randn(M, N, generator=g).cuda() if has_res else None
    wp = ops.split_pack(w)
    xs = ops.split_pack(x, 16.0).data if pre else x
    f = lambda: ops.linear_split(xs, wp, residual=r, relu=True, x_scale=16.0, x_is_split=pre)
    res = []
    for rep in range(3):
        t0, y0 = arm(0, f)
        t1, y1 = arm(1, f)
        res.append((t0, t1))
    eq = torch.equal(y0, y1)
    fl = 6.0 * M * N * K
    print(f"{name}: 128x128 {min(a for a, _ in res):.3f} ms ({fl / min(a for a, _ in res) / 1e9:.0f} TF f16)   256x128 "
          f"{min(b for _, b in res):.3f} ms ({fl / min(b for _, b in res) / 1e9:.0f} TF f16)   all {['%.3f/%.3f' % p for p in res]}  equal {eq}", flush=True)
    del x, w, r, xs, y0, y1
# the Winograd-domain batched launch
B, M, N, K = 121, R, 512, 512
x = torch.randn(B, M, K, generator=g).cuda(); w = (torch.randn(B, N, K, generator=g) * 0.05).cuda()
ws = ops.split_pack(w)
f = lambda: ops.gemm_nt_batched_split(x, ws)
res = []
for rep in range(3):
    t0, y0 = arm(0, f); t1, y1 = arm(1, f); res.append((t0, t1))
fl = 6.0 * B * M * N * K
print(f"batched 121x{M}x512x512 converting: 128x128 {min(a for a, _ in res):.3f} ms ({fl / min(a for a, _ in res) / 1e9:.0f} TF)  256x128 "
      f"{min(b for _, b in res):.3f} ms ({fl / min(b for _, b in res) / 1e9:.0f} TF)  equal {torch.equal(y0, y1)}", flush=True)
