"""Accuracy (vs an fp64 product) and speed of the split-operand GEMM next to the fp32-MFMA GEMM at the Res5 shapes."""
import sys, torch
sys.path.insert(0, ".")
from locov_amd import ops

torch.manual_seed(0)
dev = "cuda"
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

SPEED_ONLY = len(sys.argv) > 1 and sys.argv[1] == "speed"
# accuracy on a small problem incl. tiny / large magnitudes and a ragged M
for (M, N, K, mag) in [] if SPEED_ONLY else [(1000, 512, 2048, 1.0), (777, 2048, 512, 1.0), (300, 512, 512, 1e-3), (300, 512, 512, 300.0)]:
    x = (torch.randn(M, K, device=dev) * mag).relu_()
    x[::7, ::5] *= 1e-4
    w = torch.randn(N, K, device=dev) * 0.02
    ref = x.double() @ w.double().t()
    y32 = ops.linear(x, w)
    ys = ops.linear_split(x, ops.split_pack(w), x_scale=64.0 if mag < 100 else 1.0)
    den = ref.abs().max()
    print(f"M={M} N={N} K={K} mag={mag}: fp32-mfma max err {float((y32 - ref).abs().max() / den):.2e}   "
          f"split max err {float((ys - ref).abs().max() / den):.2e}   (relative to max |y|)")
# epilogue: scale / shift / residual / relu
if SPEED_ONLY:
    R = 8000
    for (M, N, K, res) in [(49 * R, 512, 2048, False), (49 * R, 2048, 512, True)]:
        x = torch.randn(M, K, device=dev).relu_(); w = torch.randn(N, K, device=dev) * 0.02
        ws = ops.split_pack(w)
        r = torch.randn(M, N, device=dev) if res else None
        f = 2.0 * M * N * K
        ts = t(lambda: ops.linear_split(x, ws, residual=r, relu=True))
        print(f"M={M} N={N} K={K} res={res}: split {ts:.3f} ms ({f / ts / 1e9:.0f} TF-equivalent)")
    sys.exit(0)
M, N, K = 1000, 512, 1024
x = torch.randn(M, K, device=dev).relu_(); w = torch.randn(N, K, device=dev) * 0.02
sc, sh, res = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev), torch.randn(M, N, device=dev)
ref = torch.relu((x.double() @ w.double().t()) * sc.double() + sh.double() + res.double())
ys = ops.linear_split(x, ops.split_pack(w), sh, scale=sc, residual=res, relu=True)
print("epilogue max err", float((ys - ref).abs().max()))
# batched
B, M, N, K = 5, 700, 512, 512
x = torch.randn(B, M, K, device=dev) * 10; w = torch.randn(B, N, K, device=dev) * 0.05
ref = torch.bmm(x.double(), w.double().transpose(1, 2))
yb = ops.gemm_nt_batched_split(x, ops.split_pack(w))
print("batched max rel err", float((yb - ref).abs().max() / ref.abs().max()), " fp32:", float((ops.gemm_nt_batched(x, w) - ref).abs().max() / ref.abs().max()))

# speed at the Res5 shapes
R = 8000
for (M, N, K, res) in [(49 * R, 512, 2048, False), (49 * R, 2048, 512, True), (33600, 2560, 1024, False)]:
    x = torch.randn(M, K, device=dev).relu_(); w = torch.randn(N, K, device=dev) * 0.02
    ws = ops.split_pack(w)
    r = torch.randn(M, N, device=dev) if res else None
    sc, sh = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
    f = 2.0 * M * N * K
    t32 = t(lambda: ops.linear(x, w, sh, scale=sc, residual=r, relu=True))
    ts = t(lambda: ops.linear_split(x, ws, sh, scale=sc, residual=r, relu=True))
    print(f"M={M} N={N} K={K} res={res}: fp32 {t32:.3f} ms ({f / t32 / 1e9:.0f} TF)   split {ts:.3f} ms ({f / ts / 1e9:.0f} TF-equivalent)")
B, M, N, K = 121, R, 512, 512
x = torch.randn(B, M, K, device=dev); w = torch.randn(B, N, K, device=dev) * 0.05
ws = ops.split_pack(w)
f = 2.0 * B * M * N * K
t32 = t(lambda: ops.gemm_nt_batched(x, w)); ts = t(lambda: ops.gemm_nt_batched_split(x, ws))
print(f"batched 121x{M}x{N}x{K}: fp32 {t32:.3f} ms ({f / t32 / 1e9:.0f} TF)   split {ts:.3f} ms ({f / ts / 1e9:.0f} TF-equivalent)")
