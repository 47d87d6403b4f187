# developer diagnostic: SQ / L2 counter passes over the split-operand GEMM (tools/attic/pmc_split.py); PMC_SPLIT_CASE selects the shape
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PMC_SPLIT_CASE=${PMC_SPLIT_CASE:-conv1}
rm -rf gpurun_out/pmcsplit*
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmcsplit$i -o p --output-format csv -- python3 tools/attic/pmc_split.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, collections, glob, os
acc = collections.defaultdict(list)
dur = []
for f in glob.glob('gpurun_out/pmcsplit*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if os.environ.get('PMC_SPLIT_KERNEL', 'gemm_split') in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('gpurun_out/pmcsplit1/p_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if os.environ.get('PMC_SPLIT_KERNEL', 'gemm_split') in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']:
            dur.append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
print('# case', os.environ['PMC_SPLIT_CASE'], ' avg launch (profiled) %.3f ms' % (sum(dur) / max(len(dur), 1) / 1e6))
for n in sorted(acc):
    print('%-34s %16.0f' % (n, sum(acc[n]) / len(acc[n])))
PY
