set -u
cd $GRAFT_REPO_ROOT
for v in "" "CENET_RING_TILE_BATCHED=128x64" "CENET_RING_TILE_BATCHED=64x128" "CENET_RING_TILE_BATCHED=128x128" "CENET_RING_TILE_BATCHED=64x64"; do
  echo "== $v"
  env $v python bench.py --no-f32 --no-cpu-baseline --no-roofline --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['launch'])"
done
