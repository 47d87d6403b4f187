"""Developer aid: phase stamps of csrc/detect.hip's selection kernel (a -DLOCOV_DET_TRACE variant build; wall_clock64 ticks at 100 MHz).
usage: python3 tools/make_variant.py dettrace detect.hip -DLOCOV_DET_TRACE && LOCOV_HIP_LIB=tools/liblocov_dettrace.so python3 tools/det_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from locov_amd import _lib

args = bench.parse(["--no-cpu-baseline"])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wl = bench.Workload(args, dev)
for _ in range(5):
    wl.step_eval(1)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 16)()
assert raw.locov_detect_trace_read(buf) == 0
t = list(buf)
names = ["load", "barrier", "sort 1", "max coordinate", "class starts", "nms rounds", "survivor keys", "sort 2", "output"]
print(f"n = {t[9]} candidates, P = {t[10]}, kept {t[11]}")
for i, nm in enumerate(names[1:], start=1):
    print(f"  {nm:16s} {(t[i] - t[i - 1]) / 100.0:9.1f} us")
print(f"  total            {(t[8] - t[0]) / 100.0:9.1f} us")
