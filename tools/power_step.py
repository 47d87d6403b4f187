"""Developer aid: package power and shader clock (rocm-smi) sampled while bench.py's S2 step (the full head, 8 x 1000 proposals) runs
back to back, and while each of its 256x256 split-GEMM launch kinds runs alone -- is the step at the package power cap, and how many
joules does it take?  Writes a table to stdout (tools/profile_round.sh keeps it as profiles/rNN_power_step.txt).

    python3 tools/power_step.py [--seconds 4]
"""
import argparse, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from locov_amd import ops


def sample():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20)
    except Exception as e:                                     # (no rocm-smi: the probe has nothing to say)
        return None, None
    pw = re.findall(r"(?:Power|SOCKET_POWER|socket_power)[^\n]*?([0-9]+\.?[0-9]*)\s*W?", r.stdout)
    ck = re.findall(r"sclk[^\n]*?\(?([0-9]+)\s*Mhz", r.stdout, flags=re.I)
    return (float(pw[0]) if pw else None), (float(ck[0]) if ck else None)


def probe(name, fn, seconds, per="launch"):
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
    time.sleep(1.5)                                            # (the power reading is a moving average)
    P, C = [], []
    t0 = time.time()
    while time.time() - t0 < seconds:
        p, c = sample()
        if p is not None: P.append(p)
        if c is not None: C.append(c)
        time.sleep(0.3)
    stop = True
    ms = 0.0
    if th:
        th.join()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
    p = sum(P) / len(P) if P else float("nan")
    c = sum(C) / len(C) if C else float("nan")
    print(f"| {name} | {ms:.3f} | {p:.0f} | {c:.0f} | {p * ms * 1e-3:.2f} | {max(p - IDLE[0], 0.0) * ms * 1e-3:.2f} |", flush=True)
    return p


IDLE = [0.0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    a = ap.parse_args()
    args = bench.parse([])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    wl = bench.Workload(args, dev)
    print(f"# {torch.cuda.get_device_name(0)}; bench.py's default workload ({args.images} x {args.proposals} proposals, {args.classes}-class bank);")
    print("# package power / shader clock = mean of rocm-smi samples while the arm runs back to back; dynamic = minus the idle reading")
    print("| arm | ms | package W | shader MHz | J | dynamic J |")
    print("|---|---|---|---|---|---|")
    IDLE[0] = probe("idle", None, a.seconds) or 0.0
    with torch.no_grad():
        probe("S2 step: the full head (bench.py's `value`)", wl.step_s2, a.seconds)
        probe("S1 step: north_star's kernel list", wl.step_s1, a.seconds)
        R = args.images * args.proposals
        M = 49 * R
        g = torch.Generator().manual_seed(0)
        x = torch.relu(torch.randn(M, 2048, generator=g)).cuda()
        xs = ops.split_pack(x, 16.0).data
        w1 = ops.split_pack((torch.randn(512, 2048, generator=g) * 0.02).cuda())
        y2 = ops.split_pack(torch.relu(torch.randn(M, 512, generator=g)).cuda(), 16.0).data
        w3 = ops.split_pack((torch.randn(2048, 512, generator=g) * 0.05).cuda())
        probe("conv1 shape [49R,2048]x[512,2048]^T, 256x256 tile", lambda: ops.linear_split(xs, w1, relu=True, x_is_split=True), a.seconds)
        probe("conv3 shape [49R,512]x[2048,512]^T + split residual, 256x256 tile",
              lambda: ops.linear_split(y2, w3, residual=xs, relu=True, x_is_split=True, residual_is_split=True, out_split=True), a.seconds)
        del x, xs, y2
        feat = wl.features["res4"]
        probe("pooler-contract ROIAlign (bit-exact)", lambda: ops.roi_align(feat, wl.rois, 14, 1.0 / 16, 0, True), a.seconds)


if __name__ == "__main__":
    main()
