"""Developer aid: where a wave of gemm_split_kernel<ASPLIT> spends a K-tile (LOCOV_HIP_LIB=tools/liblocov_ktrace.so, built by
`python tools/make_variant.py ktrace gemm_split.hip -DLOCOV_KTRACE=1`).  Segments between the s_memtime stamps of tile_step."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops, _lib
lib = _lib.lib() if hasattr(_lib, "lib") else ctypes.CDLL(os.environ["LOCOV_HIP_LIB"])
raw = ctypes.CDLL(os.environ["LOCOV_HIP_LIB"])
NAMES = ["4+4 frag reads + 8 DMA issue", "24 MFMA (GA0 x GBx, GBy)", "12 MFMA (GA1 x GBy)", "vmcnt(0) wait", "barrier", "frag reads + 12 MFMA (GA1 x GBx)", "loop back"]
g = torch.Generator().manual_seed(0)
for name, M, N, K, has_res in (("conv3 K=512 N=2048 +res", 196000, 2048, 512, True), ("wino-like K=512 N=512", 968000, 512, 512, False),
                               ("conv1 K=2048 N=512", 392000, 512, 2048, False)):
    x = torch.relu(torch.randn(M, K, generator=g)).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).cuda()
    r = torch.randn(M, N, generator=g).cuda() if has_res else None
    wp, xs = ops.split_pack(w), ops.split_pack(x, 16.0)
    f = lambda: ops.linear_split(xs.data, wp, residual=r, relu=True, x_scale=16.0, x_is_split=True)
    for _ in range(200 if K == 2048 else 20): f()          # sustained load before the stamped launch (DVFS settles)
    buf = (ctypes.c_ulonglong * 64)()
    assert raw.locov_dbg_ktrace(buf, 1) == 0
    f()
    assert raw.locov_dbg_ktrace(buf, 1) == 0
    print(name)
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    tiles = -(-M // 128) * (N // 128)
    clk0 = buf[8] / max(buf[9], 1) * 0.1
    print(f"  launch {ms:.3f} ms = {ms * 1e-3 * clk0 * 1e9 * 512 / tiles:.0f} clk per tile and workgroup slot (512 slots, {tiles} tiles)")
    n13 = max(buf[13], 1)
    print(f"  per workgroup (wave 0): entry->first DMA {buf[10]/n13:.0f}  prologue + K-loop {buf[8]/n13:.0f}  epilogue until stores issued {buf[11]/n13:.0f}  store drain {buf[12]/n13:.0f}")
    for w_ in range(4):
        d = [buf[w_ * 16 + i] for i in range(8)]
        n = max(d[7], 1)
        tot = sum(d[:7]) / n
        clk = buf[w_ * 16 + 8] / max(buf[w_ * 16 + 9], 1) * 0.1
        print(f"  wave {w_}: in-kernel clock {clk:.2f} GHz  K-tile {tot:7.0f} clk  " + "  ".join(f"{NAMES[i]}: {d[i]/n:5.0f}" for i in range(7)))
