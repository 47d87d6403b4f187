# Collects the evidence bench.py's roofline block cites, on the GPU box (run through gpurun):
#   1. bench.py JSON line                       -> gpurun_out/prof/bench.json
#   2. rocprofv3 --kernel-trace --stats         -> gpurun_out/prof/stats/
#   3. separate --pmc passes (no trace domains) -> gpurun_out/prof/pmc_{fetch,write,mfma}/
#   6. the evaluation call alone: kernel stats  -> gpurun_out/prof/eval_stats/
#   5. package power / shader clock while the step runs (tools/power_step.py) -> gpurun_out/prof/power_step.txt
#   4. the training step: kernel stats + trace  -> gpurun_out/prof_train/ (tools/profile_train.sh), host syncs -> gpurun_out/prof/find_syncs.txt
# then tools/make_profiles.py <tag> turns them into profiles/<tag>_* (run locally).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err          # the driver's command (defaults)
ARGS="--no-cpu-baseline --skip-s1 --skip-f32-reference --skip-variants --skip-train --skip-eval --steps 3 --warmup 1"      # (only the S2 step: the kernel mix of the process is the timed region's)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 bench.py --no-cpu-baseline --skip-s1 --skip-f32-reference --skip-variants --skip-train --skip-eval --steps 5 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT -d $OUT/pmc_mfma -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2>&1
# the evaluation call alone (scopes.eval_1img: one image x 1000 proposals incl. post-processing)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/eval_stats -o s --output-format csv -- python3 bench.py --only-eval --no-cpu-baseline --steps 20 > $OUT/eval.json 2> $OUT/eval.err
for m in infer eval eval_torch_chain lsm stt; do python3 tools/find_syncs.py $m 2>/dev/null | grep -v "^/"; done > $OUT/find_syncs.txt
bash tools/profile_train.sh > $OUT/profile_train.txt 2>&1
python3 tools/train_timeline.py > $OUT/train_timeline.txt 2>/dev/null
timeout 300 python3 tools/power_step.py > $OUT/power_step.txt 2>/dev/null      # package power / clock of the step and its launch kinds
ls -R $OUT | head -40
tail -c 600 $OUT/bench.json
