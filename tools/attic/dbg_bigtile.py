"""Developer aid: 128x128 (two workgroups per CU) vs 256x256 (gemm_split_big.hip: one 8-wave workgroup per CU) tile of the split GEMM
on launches with both operands pre-split, same process, alternating arms.  LOCOV_SPLIT_BIG is read by the launcher at every call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
                                    ("wino-like K=512 N=512 presplit (121 x 8000 rows as one problem)", 121 * R, 512, 512, False, True),
                                    ("ragged M, N = 264, K = 192 +res presplit", 49 * 777 + 13, 264, 192, True, True)):
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
# the Winograd-domain batched launch inside the convolution (input transform -> 121 batched GEMMs -> output transform)
x = torch.relu(torch.randn(49 * R, 512, generator=g)).cuda()
u = ops.split_pack(ops.winograd_pack_weight((torch.randn(512, 512, 3, 3, generator=g) * 0.02).cuda()))
f = lambda: ops.winograd_conv3x3(x, u, relu=True, roi_major=True, in_roi_major=True)
res = []
for rep in range(3):
    t0, y0 = arm(0, f); t1, y1 = arm(1, f); res.append((t0, t1))
print(f"winograd conv3x3 (transforms + batched GEMM): 128x128 {min(a for a, _ in res):.3f} ms   256x256 {min(b for _, b in res):.3f} ms   equal {torch.equal(y0, y1)}", flush=True)
