# developer A/B (round 4): store policy / slice count of the even-grid pooler -- time and FETCH_SIZE / WRITE_SIZE of the 2048-channel launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab_pool
for lib in product $VARIANTS; do
  for sl in 0 16; do
    if [ $lib = product ]; then unset LOCOV_HIP_LIB; else export LOCOV_HIP_LIB=tools/liblocov_$lib.so; fi
    if [ $sl = 0 ]; then unset LOCOV_ROIALIGN_SLICES; else export LOCOV_ROIALIGN_SLICES=$sl; fi
    timeout 200 python3 tools/attic/ab_pool.py 2>&1 | tail -1
    for ctr in FETCH_SIZE WRITE_SIZE; do
      d=gpurun_out/ab_pool/pmc_${lib}_${sl}_$ctr
      rm -rf $d
      timeout 120 rocprofv3 --kernel-trace --pmc $ctr -d $d -o p --output-format csv -- python3 tools/attic/ab_pool.py > /dev/null 2>&1
      python3 - <<PY
import csv, glob
v = [float(r["Counter_Value"]) for f in glob.glob("$d/**/p_counter_collection.csv", recursive=True)
     for r in csv.DictReader(open(f)) if "roi_align_nhwc_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "$ctr"]
print("   $lib slices=$sl $ctr = %.3f GB per launch%s (%d launches)" % (1024 * sum(v) / max(len(v), 1) / 1e9 * (2 if "$ctr" == "FETCH_SIZE" else 1), " (x2 corrected)" if "$ctr" == "FETCH_SIZE" else "", len(v)))
PY
    done
  done
done | tee gpurun_out/ab_pool/result.txt
