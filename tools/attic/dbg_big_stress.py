import os, sys
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from locov_amd import ops
rng = np.random.default_rng(1)
bad = tot = 0
for it in range(150):
    M = int(rng.integers(1, 3000)); N = 8 * int(rng.integers(1, 130)); K = 64 * int(rng.integers(1, 9))
    relu, out_split, aff, use_res = (bool(rng.integers(2)) for _ in range(4))
    g = torch.Generator().manual_seed(it)
    xs = ops.split_pack(torch.relu(torch.randn(M, K, generator=g)).cuda(), 16.0).data
    wp = ops.split_pack((torch.randn(N, K, generator=g) * 0.05).cuda())
    sc = (torch.rand(N, generator=g) + 0.5).cuda() if aff else None
    sh = torch.randn(N, generator=g).cuda() if aff else None
    res = torch.randn(M, N, generator=g).cuda() if use_res else None
    os.environ["LOCOV_SPLIT_BIG"] = "0"
    ref = ops.linear_split(xs, wp, sh, scale=sc, residual=res, relu=relu, x_scale=16.0, x_is_split=True, out_split=out_split)
    os.environ["LOCOV_SPLIT_BIG"] = "1"
    for rep in range(3):
        out = ops.linear_split(xs, wp, sh, scale=sc, residual=res, relu=relu, x_scale=16.0, x_is_split=True, out_split=out_split)
        tot += 1
        if not torch.equal(out, ref):
            bad += 1
            if bad < 4: print("MISMATCH", (M, N, K, relu, out_split, aff, use_res), int((out != ref).sum()), flush=True)
print("launches", tot, "mismatching", bad)
