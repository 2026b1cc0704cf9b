set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
python -m pytest tests/test_wgrad_group.py tests/test_caller_protocol.py tests/test_segmented.py tests/test_graph_replay.py -q -m gpu -x 2>&1 | tail -80 > $O/t1.log
tail -70 $O/t1.log
python tools/wgrad_bench.py 10 2>/dev/null
