# developer A/B (round 4): tile order (NG) and cache policy of the operand DMAs of gemm_split_big_kernel -- time per launch kind and
# FETCH_SIZE of the conv3 shape, one library variant per process.  Variants: tools/make_variant.py <tag> gemm_split_big.hip -D...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab_l2
for lib in product $VARIANTS; do
  if [ $lib = product ]; then unset LOCOV_HIP_LIB; else export LOCOV_HIP_LIB=tools/liblocov_$lib.so; fi
  timeout 300 python3 tools/attic/ab_big_l2.py 2>&1 | tail -1
done | tee gpurun_out/ab_l2/times.txt
for lib in product $VARIANTS; do
  if [ $lib = product ]; then unset LOCOV_HIP_LIB; else export LOCOV_HIP_LIB=tools/liblocov_$lib.so; fi
  for case in conv3_asplit conv1_asplit; do
    export PMC_SPLIT_CASE=$case
    rm -rf gpurun_out/ab_l2/pmc_$lib_$case
    timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/ab_l2/pmc_${lib}_$case -o p --output-format csv -- python3 tools/attic/pmc_split.py > /dev/null 2>&1
    python3 - <<PY
import csv, glob
v = [float(r["Counter_Value"]) for f in glob.glob("gpurun_out/ab_l2/pmc_${lib}_$case/**/p_counter_collection.csv", recursive=True)
     for r in csv.DictReader(open(f)) if "gemm_split_big" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("$lib $case FETCH_SIZE x2 = %.3f GB per launch (%d launches; M = 196000)" % (2 * 1024 * sum(v) / max(len(v), 1) / 1e9, len(v)))
PY
  done
done | tee gpurun_out/ab_l2/fetch.txt
