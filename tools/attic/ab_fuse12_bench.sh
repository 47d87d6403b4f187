# developer A/B on the GPU box: the inference bench with conv1 + conv2 of blocks 1-2 as one call (fused input transform) or two
for v in 1 0 1 0; do
  LOCOV_RES5_FUSE12=$v timeout 300 python bench.py --steps 20 --skip-s1 --skip-variants --skip-f32-reference --no-cpu-baseline --skip-train 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FUSE12=$v', round(d['value']), round(d['ms_per_step'],3))"
done
