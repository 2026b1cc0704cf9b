set -u
cd $GRAFT_REPO_ROOT
for v in "CENET_WGRAD_GROUP_SIDE=0" "CENET_WGRAD_GROUP_SIDE=1" "CENET_WGRAD_GROUP_SIDE=0" "CENET_WGRAD_GROUP_SIDE=1"; do
  echo "== $v"
  env $v python bench.py --no-f32 --no-cpu-baseline --no-roofline --steps 40 --warmup 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['launch'], d['config'].get('launch_choice'))"
done
CENET_WGRAD_GROUP_SIDE=1 python -m pytest tests/test_graph_replay.py tests/test_wellcond.py -q -m gpu 2>&1 | tail -3
