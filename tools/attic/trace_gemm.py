"""Diagnostic: per-workgroup timeline of one GEMM launch, grouped by CU (needs tools/liblocov_trace.so)."""
import sys, os, ctypes, collections
import numpy as np, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblocov_trace.so"))
vp = ctypes.c_void_p
lib.locov_gemm_nt_f32.restype = ctypes.c_int
lib.locov_gemm_nt_f32.argtypes = [vp, ctypes.c_int64, vp, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_uint, vp]
M, N, K = 196000, 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; y = torch.empty(M, N, device="cuda")
ntile = ((M + 127) // 128) * (N // 128)
st = vp(torch.cuda.current_stream().cuda_stream)
trc = torch.zeros(ntile * 8, dtype=torch.int64, device="cuda")
for _ in range(10):
    lib.locov_gemm_nt_f32(vp(x.data_ptr()), K, vp(w.data_ptr()), vp(trc.data_ptr()), None, None, vp(y.data_ptr()), N, M, N, K, 0x2000, st)
torch.cuda.synchronize()
r = trc.cpu().numpy().reshape(ntile, 8)
t0, t1, t2, hw = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
key = ((hw >> 32) << 16) | (hw & 0xffff & ~0xf) | ((hw >> 8) & 0xf)      # (xcc, se/sh/cu bits, without wave/simd ids)
cu = collections.defaultdict(list)
for b in range(ntile):
    cu[(int(hw[b] >> 32), int((hw[b] >> 8) & 0xf), int((hw[b] >> 13) & 0x7))].append(b)      # xcc, cu_id, se_id
print("workgroups", ntile, "distinct (xcc,cu,se):", len(cu))
pro = np.median(t1 - t0); epi = np.median(t2 - t1)
print("median per workgroup: start->K-loop end %d cyc, epilogue %d cyc, total %d" % (pro, epi, np.median(t2 - t0)))
for k in list(cu)[:3]:
    bs = sorted(cu[k], key=lambda b: t0[b])
    base = t0[bs[0]]
    print("CU", k, "workgroups", len(bs))
    for b in bs[:10]:
        print("   blk %6d  start %9d  loop_end %9d  end %9d   (dur %d, epi %d = barrier1 %d + relayout %d + stores issued %d + drained %d)" % (
            b, t0[b] - base, t1[b] - base, t2[b] - base, t2[b] - t0[b], t2[b] - t1[b], r[b, 4] - t1[b], r[b, 5] - r[b, 4], r[b, 6] - r[b, 5], t2[b] - r[b, 6]))
# how far apart do co-resident workgroups start?  (sorted starts per CU; consecutive differences)
d = []
for k, bl in cu.items():
    ts = np.sort(t0[bl])
    d.extend(np.diff(ts))
d = np.array(d)
print("start-to-start gaps on a CU: median %d, 10%% %d, 90%% %d cycles" % (np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
