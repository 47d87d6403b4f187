cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/bench_roialign.py 2>&1 | grep -v amdgpu
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TA_TA_BUSY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES -d gpurun_out/ra_pmc -o p --output-format csv -- python3 tools/bench_roialign.py > /dev/null 2>&1
python3 - <<'PY'
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('gpurun_out/ra_pmc/p_counter_collection.csv')):
    if 'roi_align' in r['Kernel_Name']:
        acc[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()): print('   %-28s %16.0f' % (n, sum(v)/len(v)))
PY
