"""Turns the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/prof/) into the committed evidence files
profiles/<tag>_bench.json, <tag>_bench_kernel_stats.csv, <tag>_pmc_traffic.json, <tag>_pmc_mfma_util.json."""
import collections, csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1]
dst = lambda name: os.path.join(ROOT, "profiles", f"{tag}_{name}")


def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(os.path.join(SRC, sub, "p_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def durations(sub):
    d = collections.defaultdict(list)
    with open(os.path.join(SRC, sub, "p_kernel_trace.csv")) as f:
        for r in csv.DictReader(f):
            d[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return d


bench = json.loads(open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(dst("bench.json"), "w"), indent=1)
shutil.copy(os.path.join(SRC, "stats", "s_kernel_stats.csv"), dst("bench_kernel_stats.csv"))
cfg = bench["config"]
workload = {"images": cfg["images_per_gpu"], "proposals": cfg["proposals_per_image"], "classes": cfg["classes"],
            "dim": cfg["emb_dim"], "res5": cfg["res5_backend"], "conv3x3": cfg["res5_conv3x3"],
            "block0": cfg.get("res5_block0", "pooled"), "res5_dtype": cfg.get("res5_dtype", "fp32")}

fetch, write = counters("pmc_fetch"), counters("pmc_write")
traffic = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and a separate --pmc WRITE_SIZE pass over `python3 bench.py "
                   "--no-cpu-baseline --steps 3 --warmup 1` (tools/profile_round.sh). Counter units are KiB. Correction per "
                   "MI355X_MICROARCH.md (HBM): FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads on gfx950 -> doubled; "
                   "WRITE_SIZE exact. hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch, averaged over the kernel's "
                   "launches (memory side of L2: includes Infinity-Cache hits).",
           "workload": workload, "kernels": {}}
sys.path.insert(0, ROOT)
from locov_amd import build as _build
traffic["source_fingerprint"] = _build.source_fingerprint()      # bench.py refuses this file for a library built from other sources
for k in fetch:
    if "locov" not in k:
        continue
    fs, ws = fetch[k]["FETCH_SIZE"], write.get(k, {}).get("WRITE_SIZE", [0.0])
    f_avg, w_avg = sum(fs) / len(fs), sum(ws) / len(ws)
    traffic["kernels"][k[:100]] = {"launches_sampled": len(fs), "FETCH_SIZE_KiB_avg": f_avg, "WRITE_SIZE_KiB_avg": w_avg,
                                   "hbm_bytes_per_launch": (2 * f_avg + w_avg) * 1024}
json.dump(traffic, open(dst("pmc_traffic.json"), "w"), indent=1)

m, dur = counters("pmc_mfma"), durations("pmc_mfma")
util = {"_how": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY "
                "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT over the same command (counters-only pass). mfma_util = "
                "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE/8): fraction of SIMD-cycles in which the matrix pipe is "
                "busy (GRBM_GUI_ACTIVE is summed over the 8 XCDs).", "workload": workload, "kernels": {}}
for k, c in m.items():
    if "gemm_nt_kernel" not in k and "gemm_split" not in k:
        continue
    if "pack" in k:
        continue
    avg = lambda n: sum(c[n]) / len(c[n])
    gui = avg("GRBM_GUI_ACTIVE") / 8.0
    ms = sum(dur[k]) / len(dur[k]) / 1e6
    util["kernels"][k[:100]] = {"launches": len(c["GRBM_GUI_ACTIVE"]), "mfma_util": avg("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * gui),
                                "clock_GHz_profiled": gui / (ms * 1e6), "avg_ms_profiled": ms,
                                "SQ_INSTS_MFMA": avg("SQ_INSTS_MFMA"), "SQ_INSTS_VALU": avg("SQ_INSTS_VALU"),
                                "SQ_LDS_BANK_CONFLICT": avg("SQ_LDS_BANK_CONFLICT"),
                                "SQ_WAIT_ANY_over_WAVE_CYCLES": avg("SQ_WAIT_ANY") / avg("SQ_WAVE_CYCLES")}
json.dump(util, open(dst("pmc_mfma_util.json"), "w"), indent=1)
extra = []
for src, name in ((os.path.join(SRC, "find_syncs.txt"), "find_syncs.txt"), (os.path.join(SRC, "train_timeline.txt"), "train_timeline.txt"),
                  (os.path.join(SRC, "power_step.txt"), "power_step.txt"),
                  (os.path.join(SRC, "eval_stats", "s_kernel_stats.csv"), "eval_kernel_stats.csv"),
                  (os.path.join(SRC, "eval.json"), "eval.json"),
                  (os.path.join(ROOT, "gpurun_out", "prof_train", "stats", "s_kernel_stats.csv"), "train_kernel_stats.csv")):
    if os.path.exists(src):
        shutil.copy(src, dst(name))
        extra.append(os.path.basename(dst(name)))
print("wrote", [os.path.basename(dst(n)) for n in ("bench.json", "bench_kernel_stats.csv", "pmc_traffic.json", "pmc_mfma_util.json")] + extra)
