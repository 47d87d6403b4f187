import sys, torch
sys.path.insert(0, ".")
from locov_amd import ops
torch.manual_seed(0)
for (R, N, K) in [(5, 128, 64), (300, 516, 160), (8000, 2048, 512)]:
    seg = 49; M = R * seg
    x = torch.randn(M, K, device="cuda").relu_()          # ROI-major rows
    w = torch.randn(N, K, device="cuda") * 0.05
    res_pm = torch.randn(M, N, device="cuda")              # position-major rows
    sc, sh = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
    # reference: unfused
    res_rm = res_pm.view(seg, R, N).permute(1, 0, 2).reshape(M, N)     # -> ROI-major
    full = ops.linear_split(x, ops.split_pack(w), sh, scale=sc, residual=res_rm.contiguous(), relu=True)
    want = full.view(R, seg, N).mean(dim=1)
    got = ops.linear_split_segmean(x, ops.split_pack(w), sh, res_pm, seg, scale=sc, relu=True)
    ref64 = torch.relu((x.double() @ w.double().t()) * sc.double() + sh.double() + res_rm.double()).view(R, seg, N).mean(dim=1)
    print(R, N, K, "vs unfused", float((got - want).abs().max()), "vs fp64", float((got.double() - ref64).abs().max()), "unfused vs fp64", float((want.double() - ref64).abs().max()))
    got2 = ops.linear_split_segmean(x, ops.split_pack(w), sh, res_pm, seg, scale=sc, relu=True)
    print("   deterministic:", torch.equal(got, got2))
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
ws = ops.split_pack(w)
print("fused, ROI-major residual %.3f ms" % t(lambda: ops.linear_split_segmean(x, ws, sh, res_pm, seg, scale=sc, residual_roi_major=True)))
print("fused %.3f ms   unfused GEMM %.3f ms + mean %.3f ms" % (t(lambda: ops.linear_split_segmean(x, ws, sh, res_pm, seg, scale=sc)),
      t(lambda: ops.linear_split(x, ws, sh, scale=sc, residual=res_pm, relu=True)), t(lambda: ops.spatial_mean(full.view(seg, R, N), channels_last=2) if hasattr(ops, "spatial_mean") else None)))
