"""Developer A/B: the Winograd-domain 3x3 (input transform -> 121 batched split GEMMs -> output transform) over 8 000 ROIs in ONE
pass against the same work in chunks of R ROIs whose transform-domain tensors (V, M: 31 KB per ROI and channel block each) fit the
256 MB memory-side cache.  Also the conv3-shaped split GEMM in row chunks.  HBM bytes cost ~150 pJ each here (profiles/r03_*)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from locov_amd import ops

RT = 8000
C = N = 512
g = torch.Generator().manual_seed(0)
w = torch.randn(N, C, 3, 3, generator=g) * 0.02
U = ops.split_pack(ops.winograd_pack_weight(w.cuda()))
sc, sh = (torch.rand(N, generator=g) + 0.5).cuda(), (torch.randn(N, generator=g) * 0.1).cuda()


def timed(f, n=6):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for R in (8000, 4000, 2000, 1000, 500, 250):
    xs = [torch.relu(torch.randn(49 * R, C, generator=g)).cuda() for _ in range(RT // R)]
    outs = [torch.empty(49 * R, N, device="cuda") for _ in xs]
    def f():
        for x, o in zip(xs, outs):
            ops.winograd_conv3x3(x, U, scale=sc, shift=sh, relu=True, out=o, out_split_scale=16.0)
    print(f"winograd conv2, {RT // R:3d} chunks of {R:5d} ROIs: {timed(f):.3f} ms per 8000 ROIs", flush=True)
    del xs, outs
