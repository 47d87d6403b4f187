"""Developer check of the GEMM epilogue paths (flag 0x800 forces the general path)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import _lib
lib = _lib.load()
vp = ctypes.c_void_p
torch.manual_seed(0)
for (M, N, K) in [(256, 128, 64), (1000, 768, 2048)]:
    x = torch.randn(M, K).cuda(); w = (torch.randn(N, K) * 0.05).cuda()
    ref = x.cpu().double() @ w.cpu().double().t()
    for flags in (0x800, 0):
        y = torch.full((M, N), 123.0).cuda()
        st = vp(torch.cuda.current_stream().cuda_stream)
        rc = lib.locov_gemm_nt_f32(vp(x.data_ptr()), K, vp(w.data_ptr()), None, None, None, vp(y.data_ptr()), N, M, N, K, flags, st)
        torch.cuda.synchronize()
        bad = ((y.cpu().double() - ref).abs() > 1e-3)
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print(M, N, K, hex(flags), 'rc', rc, 'bad', int(bad.sum()), 'rows', rows[:12].tolist(), 'cols', cols[:12].tolist())
        if flags == 0 and bad.any():
            r, c = bad.nonzero()[0].tolist()
            print('   first bad', r, c, 'got', float(y[r, c]), 'want', float(ref[r, c]), ' neighbours got', y[r, c-1:c+3].tolist(), 'want', ref[r, c-1:c+3].tolist())
