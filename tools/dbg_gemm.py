"""A/B timing of GEMM kernel variants (debug flags), interleaved rounds in one process after a long warm-up."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from locov_amd import _lib
lib = _lib.load()

def make(M, N, K, flags, res=False, torch_ref=False):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; y = torch.empty(M, N, device="cuda")
    r = torch.randn(M, N, device="cuda") if res else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if torch_ref:
        return lambda: torch.nn.functional.linear(x, w)
    return lambda: lib.locov_gemm_nt_f32(ctypes.c_void_p(x.data_ptr()), K, ctypes.c_void_p(w.data_ptr()), None, None,
                                         ctypes.c_void_p(r.data_ptr()) if res else None, ctypes.c_void_p(y.data_ptr()),
                                         N, M, N, K, flags, st)

def t(f, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

M = 196000
variants = [("vec", 0), ("scalar", 0x800), ("vec+stag6", 0x400 | (6 << 16)), ("hipblaslt", None)]
for (N, K, res) in [(512, 1024, False), (512, 2048, False), (2048, 512, True), (2048, 1024, False)]:
    fs = [(nm, make(M, N, K, fl or 0, res, torch_ref=fl is None)) for nm, fl in variants]
    for _ in range(30):
        for _, f in fs: f()
    torch.cuda.synchronize()
    times = {nm: [] for nm, _ in fs}
    for rnd in range(6):
        for nm, f in fs:
            times[nm].append(t(f))
    fl = 2.0 * M * N * K
    print(N, K, "res" if res else "", " ".join("%s:%.3f/%.0fTF" % (nm, np.median(v), fl / np.median(v) / 1e9) for nm, v in times.items()))
