# developer A/B (round 5): the LSM training step under two values of ONE environment knob, alternating processes on one box
# usage: bash tools/ab_train_env.sh LOCOV_RES5_BWD_STREAMS 0 1 [lsm|stt]
cd $GRAFT_REPO_ROOT
VAR=$1; A=$2; B=$3; CFG=${4:-lsm}
for rep in 1 2 3; do
  for v in $A $B; do
    echo -n "$VAR=$v $CFG: "; env $VAR=$v python3 tools/train_step_only.py --steps 60 --warmup 8 --train-config $CFG 2>/dev/null | tail -1
  done
done
