# developer A/B (round 4): the LSM / STT training step with the range guard of the training forward deferred (default) or read inside
# the step ("sync", round 3's behaviour), alternating processes on one box
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for mode in deferred sync; do
    for cfg in lsm stt; do
      echo -n "$mode $cfg: "; LOCOV_RES5_TRAIN_GUARD=$mode python3 tools/train_step_only.py --steps 60 --warmup 8 --train-config $cfg 2>/dev/null | tail -1
    done
  done
done
