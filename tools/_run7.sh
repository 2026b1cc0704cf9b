set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
python -m pytest tests/test_segmented.py tests/test_graph_replay.py tests/test_bench_dist.py -q -m gpu 2>&1 | tail -30 > $O/t1.log
tail -8 $O/t1.log
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python3 $GRAFT_REPO_ROOT/bench.py --config ham512 --no-cpu-baseline --no-f32 --no-roofline --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$O/ham.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $O/ham.log | cut -c1-400
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/ham_kernel_stats.csv
rm -rf $O/stats
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3g/ham_kernel_stats.csv')))
S=[int(r['Calls']) for r in rows if r['Name'].startswith('sgd_kernel')][0]
tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6/S
print('steps',S,'ms/step',tot)
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/1e6/S:7.3f} ms  {int(r['Calls'])/S:6.1f} x {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:95]}")
PY
