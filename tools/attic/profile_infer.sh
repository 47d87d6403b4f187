# rocprofv3 kernel stats of the inference step (bench.py defaults, S2 only) on the GPU box -> gpurun_out/prof_infer/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_infer
rm -rf $OUT && mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 bench.py --no-cpu-baseline --skip-s1 --skip-f32-reference --skip-variants --steps 8 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel ms per step", tot / 1e6 / 10)
for r in rows[:14]:
    print(f'{float(r["TotalDurationNs"])/1e6/10:9.3f} ms/step {int(r["Calls"])/10:6.1f} calls/step {float(r["AverageNs"])/1e3:9.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%  {r["Name"][:100]}')
PY
