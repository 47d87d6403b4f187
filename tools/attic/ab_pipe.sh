# developer A/B (round 4): the pipelined pooler-contract ROIAlign (roi_align_pipe.hip) against the one-workgroup-per-(ROI, 32 channels)
# form (LOCOV_ROIALIGN_PIPE=0) and its own variants; then the WINO pooler's store policy end to end
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo -n "T2 (pipe off): "; LOCOV_ROIALIGN_PIPE=0 python3 tools/ab_t2.py 2>/dev/null | tail -1
  for lib in product $VARIANTS; do
    if [ $lib = product ]; then unset LOCOV_HIP_LIB; else export LOCOV_HIP_LIB=tools/liblocov_$lib.so; fi
    python3 tools/ab_t2.py 2>/dev/null | tail -1
  done
  unset LOCOV_HIP_LIB
done
for rep in 1 2; do
  for lib in product poolwino_nt; do
    if [ $lib = product ]; then unset LOCOV_HIP_LIB; else export LOCOV_HIP_LIB=tools/liblocov_$lib.so; fi
    echo -n "$lib S2: "; python3 bench.py --no-cpu-baseline --skip-s1 --skip-f32-reference --skip-variants --skip-train --steps 20 --warmup 5 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['value'], b['ms_per_step'])"
  done
  unset LOCOV_HIP_LIB
done
