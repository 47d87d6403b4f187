# texture-address / L1 / VALU counters of the pooler-contract ROIAlign (and the even-grid pooler) at the bench workload -> gpurun_out/ra_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/ab_t2.py 2>/dev/null | tail -1 > gpurun_out/ra_pmc.txt
python3 tools/attic/ab_pool.py 2>/dev/null | tail -1 >> gpurun_out/ra_pmc.txt
for tgt in ab_t2 ab_pool; do
  rm -rf gpurun_out/ra_pmc_$tgt
  timeout 200 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TA_TA_BUSY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/ra_pmc_$tgt -o p --output-format csv -- python3 tools/$tgt.py > /dev/null 2>&1
done
python3 - >> gpurun_out/ra_pmc.txt <<'PY'
import csv, collections, glob
for tgt in ("ab_t2", "ab_pool"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/ra_pmc_{tgt}/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "roi_align" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"gpurun_out/ra_pmc_{tgt}/**/p_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "roi_align" in r["Kernel_Name"]:
                dur[r["Kernel_Name"][:60]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, c in acc.items():
        ms = sum(dur[k]) / max(len(dur[k]), 1) / 1e6
        print(f"{k}   ({len(dur[k])} launches, {ms:.3f} ms profiled)")
        for n, v in sorted(c.items()):
            print("   %-28s %16.0f" % (n, sum(v) / len(v)))
        if "GRBM_GUI_ACTIVE" in c and "TA_TA_BUSY" in c:
            cyc = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / 8.0
            print("   -> TA busy / (256 CUs x cycles) = %.2f; L1 accesses x 64 B = %.1f GB, L1 -> L2 read requests x 128 B = %.1f GB"
                  % (sum(c["TA_TA_BUSY"]) / len(c["TA_TA_BUSY"]) / (256.0 * cyc), sum(c["TCP_TOTAL_CACHE_ACCESSES"]) / len(c["TCP_TOTAL_CACHE_ACCESSES"]) * 64 / 1e9,
                     sum(c["TCP_TCC_READ_REQ"]) / len(c["TCP_TCC_READ_REQ"]) * 128 / 1e9))
PY
cat gpurun_out/ra_pmc.txt
