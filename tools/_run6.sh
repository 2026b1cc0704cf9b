set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
python -m pytest tests/test_wgrad_group.py tests/test_caller_protocol.py tests/test_segmented.py tests/test_graph_replay.py -q -m gpu 2>&1 | tail -80 > $O/t1.log
tail -50 $O/t1.log
python tools/wgrad_bench.py 10 2>/dev/null
for v in "" "CENET_GROUP_ITEMS=1024" "CENET_GROUP_ITEMS=2560"; do
  echo "== bench $v"
  env $v python bench.py --no-f32 --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['launch'])
for k in d['roofline']['top_kernels']: print('   ',k['kernel'],k['launches_per_step'],k['avg_launch_ms'],k['total_ms_per_step'],k.get('frac'))
print('   ', {k:v for k,v in d['roofline']['next_kernels_ms_per_step'].items()})
"
done
