"""Times the ROIAlign kernels at the bench workload (developer tool)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from locov_amd import ops, _lib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
lib = _lib.load()
dev = torch.device("cuda")
gen = torch.Generator().manual_seed(1992)
B, R = 4, 1000
feat = torch.randn(B, 1024, 50, 84, generator=gen).to(dev)
rois = bench.synth_rois(gen, B, R, dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
out = torch.empty(B * R, 1024, 14, 14, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
def old(): lib.locov_roi_align_fwd(p(feat), B, 1024, 50, 84, p(rois), B * R, 14, 14, 1 / 16, 0, 1, p(out), st)
def new(): ops.roi_align(feat, rois, 14, 1 / 16, 0, True)
nhwc = ops.nchw_to_nhwc(feat)
def even(): ops.roi_align_nhwc(nhwc, rois, 14, 1 / 16, 0, True, bin_stride=2, pos_major=True)
def t(f, n=10):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
gb = (B * 1024 * 50 * 84 * 4 + B * R * 1024 * 196 * 4) / 1e9
for nm, f in (("nchw gather (old)", old), ("nhwc gather + LDS transpose (incl. nhwc copy)", new)):
    ms = t(f); print(f"{nm}: {ms:.3f} ms  {gb / ms:.2f} TB/s algorithmic")
ms = t(even); print(f"even-grid nhwc position-major: {ms:.3f} ms  {(B*R*49*1024*4/1e9) / ms:.2f} TB/s of output")
a = out.clone(); old(); torch.cuda.synchronize()
print("bit-identical:", torch.equal(out, ops.roi_align(feat, rois, 14, 1 / 16, 0, True)))
