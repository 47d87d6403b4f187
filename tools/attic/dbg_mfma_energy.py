"""Developer aid: runs tools/probe/mfma_energy_probe (built on the spot) as a child and samples package power / shader clock beside it."""
import os, re, subprocess, sys, time
here = os.path.dirname(os.path.abspath(__file__))
exe = "/tmp/mfma_energy_probe"
subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", os.path.join(here, "probe", "mfma_energy_probe.hip"), "-o", exe])
secs = sys.argv[1] if len(sys.argv) > 1 else "5"
child = subprocess.Popen([exe, secs], stdout=subprocess.PIPE, text=True)
t0 = time.monotonic()
while child.poll() is None:
    r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
    pw = re.findall(r"Power[^\n]*?([0-9]+\.?[0-9]*)\s*$", r, flags=re.M)
    ck = re.findall(r"sclk[^\n]*?\(([0-9]+)Mhz", r, flags=re.I)
    print(f"t+{time.monotonic() - t0:5.1f} s  power {pw[:1]} W  sclk {ck[:1]} MHz", flush=True)
    time.sleep(0.7)
print(child.stdout.read())
