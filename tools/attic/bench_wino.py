"""Developer timing: direct (tap-skipping implicit GEMM) vs Winograd-domain 3x3 conv at the Res5 shape."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from locov_amd import ops

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
C = N = 512
x = torch.randn(49 * R, C, device="cuda")
w = torch.randn(N, C, 3, 3, device="cuda") * 0.02
sc, sh = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda") * 0.1
wp, U = ops.pack_conv3x3_weight(w), ops.winograd_pack_weight(w)
fd = lambda: ops.conv3x3_nhwc(x, wp, 7, 7, scale=sc, shift=sh, relu=True, pos_major=True)
fw = lambda: ops.winograd_conv3x3(x, U, scale=sc, shift=sh, relu=True)
a, b = fd(), fw()
print("max |direct - winograd| =", (a - b).abs().max().item(), " max |y| =", a.abs().max().item())
for _ in range(5):
    fd(); fw()
for name, f in (("direct", fd), ("winograd", fw), ("direct", fd), ("winograd", fw)):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:9s} {e0.elapsed_time(e1) / 10:.3f} ms")
