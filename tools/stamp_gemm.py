"""Diagnostic build only: per-phase s_memtime shares of the GEMM K-loop (workgroup 300, wave 0)."""
import sys, os, ctypes
import numpy as np, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblocov_stamp.so"))
lib.locov_gemm_nt_f32.restype = ctypes.c_int
vp = ctypes.c_void_p
lib.locov_gemm_nt_f32.argtypes = [vp, ctypes.c_int64, vp, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_uint, vp]
M, N, K = 196000, 512, 2048
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; y = torch.empty(M, N, device="cuda")
st = vp(torch.cuda.current_stream().cuda_stream)
names = ["A(16 mfma)", "B(16 mfma)", "C(stage+16 mfma)", "D1(8 mfma)", "barrier", "D2(read+8 mfma)"]
for label, extra in (("normal", 0), ("no ds_write in staging", 0x10000), ("no refill loads in staging", 0x20000), ("neither", 0x30000)):
    dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
    for it in range(20):
        assert lib.locov_gemm_nt_f32(vp(x.data_ptr()), K, vp(w.data_ptr()), vp(dbg.data_ptr()), None, None, vp(y.data_ptr()), N, M, N, K, 0x2000 | extra, st) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(5):
        lib.locov_gemm_nt_f32(vp(x.data_ptr()), K, vp(w.data_ptr()), vp(dbg.data_ptr()), None, None, vp(y.data_ptr()), N, M, N, K, 0x2000 | extra, st)
    e1.record(); torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.float64) / (K // 32 - 1)
    r = dbg.cpu().numpy().astype(np.float64)
    clk = (r[12] - r[8]) / (r[13] - r[9]) * 0.1
    print("   block 300: prologue %.0f cyc, K-loop %.0f cyc, epilogue %.0f cyc, total %.0f cyc; in-kernel clock %.3f GHz" % (r[10] - r[8], r[11] - r[10], r[12] - r[11], r[12] - r[8], clk))
    print("cfg", os.environ.get("LOCOV_GEMM_CFG"), label, "%.3f ms" % (e0.elapsed_time(e1) / 5), " ".join("%s=%.0f" % (n.split("(")[0], v) for n, v in zip(names, d[:6])), "total=%.0f" % d[:6].sum())
