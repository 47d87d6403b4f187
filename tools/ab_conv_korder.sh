cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
 python bench.py --no-cpu-baseline --steps 10 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chan-outer', d['value'], d['roofline']['avg_launch_ms'])"
 LOCOV_HIP_LIB=$GRAFT_REPO_ROOT/tools/liblocov_tapouter.so python bench.py --no-cpu-baseline --steps 10 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tap-outer', d['value'], d['roofline']['avg_launch_ms'])"
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
ls gpurun_out/pmc_fetch
