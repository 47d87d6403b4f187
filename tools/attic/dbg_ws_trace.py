"""Developer aid: barrier-wait cycles per role of the wave-specialised split GEMM (LOCOV_HIP_LIB=tools/liblocov_wsdbg.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops, _lib
g = torch.Generator().manual_seed(0)
for name, M, N, K, has_res in (("k2048", 70001, 512, 2048, False), ("k512_res", 70000, 2048, 512, True)):
    x = torch.relu(torch.randn(M, K, generator=g)).cuda()
    wp = ops.split_pack((torch.randn(N, K, generator=g) * 0.05).cuda())
    xs = ops.split_pack(x, 16.0)
    r = torch.randn(M, N, generator=g).cuda() if has_res else None
    for _ in range(3):
        y = ops.linear_split(xs.data, wp, residual=r, relu=True, x_scale=16.0, x_is_split=True)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    lib = _lib.load()
    lib.locov_ws_debug_read.argtypes = [ctypes.c_void_p]
    assert lib.locov_ws_debug_read(buf) == 0
    KT = K // 32
    tiles = ((M + 127) // 128) * (N // 128)
    T = (tiles - 8 + 255) // 256
    for role, nm in enumerate(("mfma", "staging", "epilogue")):
        print(f"{name} {nm:9s}: barrier wait {buf[2*role]:>10d} of {buf[2*role+1]:>10d} cycles ({100.0*buf[2*role]/max(buf[2*role+1],1):.0f}%), per K-tile total {buf[2*role+1]/(T*KT):.0f} cyc")
