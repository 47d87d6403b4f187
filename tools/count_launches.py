"""Developer aid: kernel launches of one call, counted with torch.profiler (kineto) -- does it work on this ROCm build, and what
does it say for the evaluation step?  usage: python3 tools/count_launches.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def count(fn, warm=3):
    from torch.profiler import profile, ProfilerActivity
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        fn()
        torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    names = {}
    for e in ev:
        names[e.name] = names.get(e.name, 0) + 1
    return len(ev), names


if __name__ == "__main__":
    args = bench.parse(["--no-cpu-baseline"])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    wl = bench.Workload(args, dev)
    t0 = time.time()
    n, names = count(wl.step_eval)
    print(f"eval_1img: {n} device events in {time.time() - t0:.1f} s")
    for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:40]:
        print(f"  {v:4d}  {k[:120]}")
    n8, _ = count(lambda: wl.step_s2())
    print(f"S2 step (8 images, no post-processing): {n8} device events")
