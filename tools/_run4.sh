set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
python -m pytest tests/test_caller_protocol.py tests/test_segmented.py tests/test_graph_replay.py tests/test_bench_dist.py tests/test_evaluate.py tests/test_kern_gemm.py tests/test_kern_misc.py tests/test_losses.py tests/test_merged_branches.py tests/test_optim_sched.py tests/test_overlap.py tests/test_data.py tests/test_checkpoint.py tests/test_attention_presets.py tests/test_bf16_storage.py -q -m gpu 2>&1 | tail -40 > $O/gpu_tests.log
tail -12 $O/gpu_tests.log
python tools/gemm_census.py bf16 > $O/census.log 2>&1
head -75 $O/census.log | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-f32 --no-roofline --steps 30 --warmup 5 > $GRAFT_REPO_ROOT/$O/stats.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python tools/prof_summary.py $O/stats 30 > $O/prof_summary.txt 2>&1
cat $O/prof_summary.txt
rm -rf $O/stats
