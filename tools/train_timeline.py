"""Turns the rocprofv3 kernel trace of tools/profile_train.sh (gpurun_out/prof_train/stats/s_kernel_trace.csv) into a phase timeline of
ONE training step: consecutive launches of the same kernel (and runs of sub-12 us torch launches) merged, with the time the GPU was busy
inside each run next to the run's span -- where busy << span the host could not launch fast enough (it had waited for the GPU just before).
usage: python tools/train_timeline.py [trace.csv] > profiles/<tag>_train_timeline.txt"""
import csv, glob, re, sys
f = sys.argv[1] if len(sys.argv) > 1 else glob.glob("gpurun_out/prof_train/stats/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "roi_align_even_bwd_tiles_kernel" in r["Kernel_Name"]
         or "roi_align_nhwc_kernel<float, float, true" in r["Kernel_Name"]]                                   # one ROIAlign backward per step
seg = rows[marks[-2] + 1: marks[-1] + 1]


def short(n):
    n = n.replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", "")
    return re.match(r"([A-Za-z0-9_:]+)", n).group(1)[:56]


t0 = int(seg[0]["Start_Timestamp"])
busy_all = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"# one LSM training step (the last of the profiled run): {len(seg)} launches, GPU busy {busy_all / 1e6:.2f} ms inside a span of "
      f"{(int(seg[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms (profiled: rocprofv3 adds host time per launch)")
print("#   start      run of                                                    launches      busy       span")
out, prev = [], None
for r in seg:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    s = short(r["Kernel_Name"])
    key = "(sub-12 us torch launches)" if d < 12000 and "locov" not in s else s
    if key == prev:
        out[-1][2] += 1; out[-1][3] += d; out[-1][4] = int(r["End_Timestamp"])
    else:
        out.append([int(r["Start_Timestamp"]), key, 1, d, int(r["End_Timestamp"])])
        prev = key
for ts, k, c, d, te in out:
    flag = "   <-- host-bound" if (te - ts) > 3 * d and (te - ts) > 100000 else ""
    print(f"{(ts - t0) / 1e6:8.3f} ms  {k:58s} x{c:4d}  {d / 1e3:8.1f} us  {(te - ts) / 1e3:8.1f} us{flag}")
