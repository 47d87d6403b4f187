"""Developer aid: shader clock and package power (rocm-smi / amd-smi, whichever answers) sampled while one kernel runs back to back:
is the split GEMM power-limited?  Arms: the shipped conv1 / conv3 launches, the f32-MFMA GEMM, an HBM-bound transform, idle."""
import os, subprocess, sys, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops


def sample():
    out = {}
    for cmd in (["rocm-smi", "--showpower", "--showclocks", "--showtemp"], ["amd-smi", "metric", "-p", "-c"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            out[cmd[0]] = r.stdout[-1500:] + r.stderr[-300:]
        except Exception as e:
            out[cmd[0]] = f"failed: {e}"
    return out


def grab(txt):
    pw = re.findall(r"(?:Power|SOCKET_POWER|socket_power)[^\n]*?([0-9]+\.?[0-9]*)\s*W?", txt)
    ck = re.findall(r"sclk[^\n]*?\(?([0-9]+)\s*Mhz", txt, flags=re.I)
    return pw[:2], ck[:2]


g = torch.Generator().manual_seed(0)
R = 8000
M = 49 * R
x = torch.relu(torch.randn(M, 2048, generator=g)).cuda()
xs = ops.split_pack(x, 16.0).data
w1 = ops.split_pack((torch.randn(512, 2048, generator=g) * 0.02).cuda())
w1f = (torch.randn(512, 2048, generator=g) * 0.02).cuda()
y2 = ops.split_pack(torch.relu(torch.randn(M, 512, generator=g)).cuda(), 16.0).data
w3 = ops.split_pack((torch.randn(2048, 512, generator=g) * 0.05).cuda())
only = os.environ.get("ARMS")
arms = {
    "idle": None,
    "conv1 presplit (K=2048)": lambda: ops.linear_split(xs, w1, relu=True, x_is_split=True),
    "conv1 converting": lambda: ops.linear_split(x, w1, relu=True),
    "conv3 presplit + residual (K=512)": lambda: ops.linear_split(y2, w3, residual=x, relu=True, x_is_split=True),
    "f32 MFMA GEMM (K=2048)": lambda: ops.linear(x, w1f, relu=True),
    "nchw->nhwc copy (HBM-bound)": lambda: ops.nchw_to_nhwc(x.view(1960, 2048, 10, 20)),
}
if only:
    arms = {k: v for k, v in arms.items() if any(o in k for o in only.split(","))}
for name, fn in arms.items():
    stop = False

    def loop():
        while not stop:
            fn()
        torch.cuda.synchronize()

    th = None
    if fn is not None:
        for _ in range(3): fn()
        torch.cuda.synchronize()
        th = threading.Thread(target=loop); th.start()
    time.sleep(2.0)
    got = []
    for _ in range(3):
        s = sample()
        got.append({k: grab(v) for k, v in s.items()})
        time.sleep(0.5)
    stop = True
    if th: th.join()
    if fn is not None:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
    else:
        ms = 0.0
    pw = [float(x) for s_ in got for x in s_["rocm-smi"][0][:1]]
    ck = [float(x) for s_ in got for x in s_["rocm-smi"][1][:1]]
    P, C = (sum(pw) / len(pw) if pw else 0.0), (sum(ck) / len(ck) if ck else 0.0)
    print(f"{os.environ.get('LOCOV_HIP_LIB', 'product')[-20:]:>20s} | {name}: {ms:.3f} ms  {P:.0f} W  {C:.0f} MHz  {P * ms * 1e-3:.3f} J/launch", flush=True)
