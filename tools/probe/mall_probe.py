"""Developer probe: does a producer -> consumer pair whose tensor fits the 256 MB memory-side cache (Infinity Cache) run faster / cheaper than
one that goes through HBM?  Ping-pong copy a -> b, b -> a of S bytes (each pass reads what the previous one wrote), bandwidth and package
power per working-set size."""
import os, re, subprocess, sys, threading, time
import torch


def power():
    r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20)
    pw = re.findall(r"(?:Power|SOCKET_POWER|socket_power)[^\n]*?([0-9]+\.?[0-9]*)\s*W?", r.stdout)
    ck = re.findall(r"sclk[^\n]*?\(?([0-9]+)\s*Mhz", r.stdout, flags=re.I)
    return float(pw[0]) if pw else float("nan"), float(ck[0]) if ck else float("nan")


print("| tensor MB | copy GB/s (read + write) | package W | pJ per byte moved (dynamic, idle subtracted) |")
print("|---|---|---|---|")
idle = sum(power()[0] for _ in range(3)) / 3
for mb in (16, 32, 64, 96, 128, 192, 256, 512, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.randn(n, device="cuda"); b = torch.empty_like(a)
    reps = max(4, int(8e9 / (mb * 2 ** 20)))
    stop = False

    def loop():
        while not stop:
            for _ in range(reps):
                b.copy_(a); a.copy_(b)
        torch.cuda.synchronize()

    for _ in range(3): b.copy_(a); a.copy_(b)
    torch.cuda.synchronize()
    th = threading.Thread(target=loop); th.start()
    time.sleep(1.5)
    P = [power()[0] for _ in range(6)]
    stop = True; th.join()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): b.copy_(a); a.copy_(b)
    e1.record(); torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3
    moved = 2 * reps * 2 * mb * 2 ** 20
    p = sum(P) / len(P)
    print(f"| {mb} | {moved / sec / 1e9:.0f} | {p:.0f} | {(p - idle) * sec / moved * 1e12:.0f} |", flush=True)
    del a, b
print(f"# idle {idle:.0f} W")
