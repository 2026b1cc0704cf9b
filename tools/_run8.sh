set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
python -m pytest tests/test_attention_presets.py tests/test_sra_blocks.py tests/test_bf16_storage.py tests/test_segmented.py tests/test_graph_replay.py tests/test_bench_dist.py tests/test_ham_oracle.py -q -m gpu 2>&1 | tail -30 > $O/t1.log
tail -8 $O/t1.log
python bench.py --config ham512 --no-cpu-baseline --no-f32 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ham512', d['value'], d['ms_per_step'], d['config']['launch'])
for k in d['roofline']['top_kernels']: print('   ',k['kernel'],k['launches_per_step'],k['avg_launch_ms'],k['total_ms_per_step'],k.get('frac'))
print('   ', d['roofline']['next_kernels_ms_per_step'])
"
python bench.py --no-cpu-baseline --no-f32 --no-roofline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('acdc', d['value'], d['ms_per_step'], d['config']['launch'])"
