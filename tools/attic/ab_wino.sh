# developer A/B of the Winograd transforms (build variants with tools/make_variant.py <tag> winograd.hip -DLOCOV_WINO_NT_STORE=1 ..., AB_LIBS="<tag> ..."): per-kernel times from rocprofv3 kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in product ${AB_LIBS}; do
  if [ $lib = product ]; then unset LOCOV_HIP_LIB; else export LOCOV_HIP_LIB=tools/liblocov_$lib.so; fi
  rm -rf gpurun_out/abwino; timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/abwino -o s --output-format csv -- python3 tools/attic/bench_wino_transforms.py > /dev/null 2>&1
  echo "== $lib"
  python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/abwino/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print("  %-60s %4s calls %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
