# developer A/B: conv tile-group size (LOCOV_CONV_GROUP) -> throughput, conv launch time, L2-miss traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
IM=${IMAGES:-4}
for rep in 1 2; do
for g in 1 2 4 8 16 1000; do
 LOCOV_CONV_GROUP=$g python bench.py --no-cpu-baseline --images $IM --steps 10 --warmup 5 2>/dev/null | G=$g python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('G', os.environ['G'], round(d['value']), round(d['roofline']['avg_launch_ms'],3))"
done; done
for g in 1 4 16; do
 export LOCOV_CONV_GROUP=$g
 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_g$g -o f --output-format csv -- python3 bench.py --no-cpu-baseline --images $IM --steps 3 --warmup 1 > /dev/null 2>&1
 python - <<PY
import csv
v=[float(r['Counter_Value']) for r in csv.DictReader(open('gpurun_out/pmc_g$g/f_counter_collection.csv')) if r['Counter_Name']=='FETCH_SIZE' and '2, 2, 2, 2, false' in r['Kernel_Name']]
print('G $g conv FETCH GB', round(sum(v)/len(v)*2048/1e9,2), len(v))
PY
done
