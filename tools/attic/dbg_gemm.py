"""Per-shape timing of the hand-written NT GEMM against hipBLASLt (torch) at the Res5 shapes, interleaved rounds in
one process after a long warm-up (first-run timings are clock-ramp biased)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from locov_amd import _lib, ops
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


def make_batched(B, M, N, K, torch_ref=False):
    x = torch.randn(B, M, K, device="cuda"); w = torch.randn(B, N, K, device="cuda") * 0.02
    if torch_ref:
        wt = w.transpose(1, 2)
        return lambda: torch.bmm(x, wt)
    return lambda: ops.gemm_nt_batched(x, w)


def t(f, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def ab(label, fl, fs):
    for _ in range(30):
        for _, f in fs: f()
    torch.cuda.synchronize()
    times = {nm: [] for nm, _ in fs}
    for rnd in range(6):
        for nm, f in fs:
            times[nm].append(t(f))
    print(label, " ".join("%s:%.3f ms/%.0f TF" % (nm, np.median(v), fl / np.median(v) / 1e9) for nm, v in times.items()), flush=True)


R = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
M = 49 * R
# (N, K, residual): conv1 of block 0, conv1 of blocks 1-2, conv3 (+residual), shortcut
for (N, K, res) in [(512, 1024, False), (512, 2048, False), (2048, 512, True), (2048, 512, False), (2048, 1024, False),
                    (2048, 1536, False)]:
    ab(f"M={M} N={N} K={K} {'res' if res else '   '}", 2.0 * M * N * K,
       [("locov", make(M, N, K, 0, res)), ("hipblaslt", make(M, N, K, 0, res, torch_ref=True))])
ab(f"batched 121 x [{R},512]x[512,512]^T", 2.0 * 121 * R * 512 * 512,
   [("locov", make_batched(121, R, 512, 512)), ("hipblaslt", make_batched(121, R, 512, 512, torch_ref=True))])
ab(f"emb_pred M={R} N=768 K=2048", 2.0 * R * 768 * 2048, [("locov", make(R, 768, 2048, 0)), ("hipblaslt", make(R, 768, 2048, 0, torch_ref=True))])
