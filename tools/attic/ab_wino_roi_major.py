import sys, torch
sys.path.insert(0, ".")
from locov_amd import ops
R, C, N = 8000, 512, 512
x = torch.randn(49 * R, C, device="cuda").relu_()
U = ops.split_pack(ops.winograd_pack_weight((torch.randn(N, C, 3, 3, device="cuda") * 0.05)))
sc, sh = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for rm in (False, True, False, True):
    print("roi_major", rm, "%.3f ms" % t(lambda: ops.winograd_conv3x3(x, U, scale=sc, shift=sh, relu=True, roi_major=rm)))
