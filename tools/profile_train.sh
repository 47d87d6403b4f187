# rocprofv3 kernel stats of the training step (bench.py --mode train) on the GPU box -> gpurun_out/prof_train/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_train
rm -rf $OUT && mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 tools/train_step_only.py --steps 6 --warmup 2 $TRAIN_ARGS > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:34]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {int(r["Calls"]):5d} calls {float(r["AverageNs"])/1e3:9.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%  {r["Name"][:110]}')
PY
