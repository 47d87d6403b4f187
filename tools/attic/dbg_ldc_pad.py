"""Developer experiment: how much do the row pitch and the base-address offset of the GEMM's output / residual matter
(L2 / HBM channel mapping of the epilogue's traffic)?  One arena allocated once; views carved at chosen offsets; every
configuration measured twice in shuffled order."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops, _lib
from locov_amd.ops import _ptr, _stream, _overflow_word, check
lib = _lib.load()
g = torch.Generator().manual_seed(0)
arena_y = torch.empty(1 << 30, dtype=torch.float32, device="cuda")       # 4 GB each
arena_r = torch.randn(1 << 29, device="cuda").repeat(2)
for name, M, N, K, has_res in (("conv3 K=512 N=2048 +res", 196000, 2048, 512, True), ("wino-like K=512 N=512", 968000, 512, 512, False)):
    x = torch.relu(torch.randn(M, K, generator=g)).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).cuda()
    wp, xs = ops.split_pack(w), ops.split_pack(x, 16.0)
    cfgs = [(pad, off) for pad in (0, 32, 96, 128) for off in (0, 64, 1024, 4096 + 64)]      # off in floats
    res = {c: [] for c in cfgs}
    for rep in range(2):
        order = cfgs[:]
        random.Random(rep).shuffle(order)
        for pad, off in order:
            ldc = N + pad
            yb = arena_y[off:off + M * ldc].view(M, ldc)
            rb = arena_r[off:off + M * ldc].view(M, ldc) if has_res else None
            def f():
                check(lib.locov_gemm_nt_f32_split(_ptr(xs.data), K, _ptr(wp.data), None, None, _ptr(rb), _ptr(yb), ldc, M, N, K,
                                                  _lib.EPI_RELU | _lib.GEMM_A_SPLIT, 16.0, wp.scale, _ptr(_overflow_word(x)), _stream(x)), "gemm")
            f(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): f()
            torch.cuda.synchronize(); res[(pad, off)].append((time.perf_counter() - t0) / 5 * 1e3)
    print(name)
    for pad in (0, 32, 96, 128):
        print(f"  pitch N + {pad:3d}: " + "   ".join(f"off {off:5d}: {res[(pad, off)][0]:.3f} / {res[(pad, off)][1]:.3f}" for off in (0, 64, 1024, 4096 + 64)), flush=True)
