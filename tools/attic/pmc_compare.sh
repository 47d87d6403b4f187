# developer diagnostic: SQ counter passes over one launch set of the hand-written GEMM and hipBLASLt (tools/attic/pmc_gemm.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmccmp$i -o p --output-format csv -- python3 tools/attic/pmc_gemm.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmccmp*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = 'locov' if 'gemm_nt_kernel' in r['Kernel_Name'] else ('hipblaslt' if r['Kernel_Name'].startswith('Cijk') else None)
        if k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted(set(acc['locov']) | set(acc['hipblaslt']))
print('%-34s %16s %16s' % ('counter (avg per launch)', 'locov', 'hipblaslt'))
for n in names:
    a = acc['locov'].get(n, [0]); b = acc['hipblaslt'].get(n, [0])
    print('%-34s %16.0f %16.0f' % (n, sum(a)/len(a), sum(b)/len(b)))
PY
