#!/bin/bash
# Regenerates, on the MI355X box, every measured artefact that profiles/README.md cites for the current round:
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh r03'
# Outputs land in gpurun_out/refresh_<tag>/ (copy the ones to be judged into profiles/).  Order matters: the PMC passes come
# first, because bench.py reads roofline.traffic from profiles/<tag>_traffic.json by kernel name.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/refresh_$TAG
mkdir -p $O
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
PMC="--steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --graph off"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $PMC > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py $PMC > $O/pmc_write.log 2>&1
F=$(ls $O/pmc_fetch/*/*counter_collection.csv | head -1); W=$(ls $O/pmc_write/*/*counter_collection.csv | head -1)
python3 $R/tools/pmc_traffic.py $F $W $R/profiles/${TAG}_traffic.json "bench.py $PMC" > $O/pmc_top.txt 2>&1
cp $R/profiles/${TAG}_traffic.json $O/
# kernel statistics of the default bench command (instrumented passes included) and of the step alone
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b -- python3 $R/bench.py --no-cpu-baseline --no-f32 > $O/stats_b.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c -- python3 $R/bench.py --no-cpu-baseline --no-f32 --no-roofline > $O/stats_c.log 2>&1
cp $(ls $O/stats_b/*/*kernel_stats.csv | head -1) $O/${TAG}_b_kernel_stats.csv
cp $(ls $O/stats_c/*/*kernel_stats.csv | head -1) $O/${TAG}_c_kernel_stats_noroofline.csv
cd $R
python3 tools/prof_summary.py $O/stats_c 25 > $O/prof_summary_c.txt 2>&1
rm -rf $O/stats_b $O/stats_c $O/pmc_fetch/*/*kernel_trace.csv $O/pmc_write/*/*kernel_trace.csv   # (traces: tens of MB)
cp $O/${TAG}_c_kernel_stats_noroofline.csv profiles/   # (bench.py's roofline.rocprof_avg_launch_ms reads the latest round's summary)
# the bench lines
python3 bench.py > $O/${TAG}_bench_50steps.json 2> $O/bench.err
python3 bench.py --config synapse --no-f32 --no-cpu-baseline > $O/${TAG}_bench_synapse.json 2>> $O/bench.err
python3 bench.py --config ham512 --no-f32 --no-cpu-baseline --steps 20 --warmup 5 > $O/${TAG}_bench_ham512.json 2>> $O/bench.err
# attention kernels: SQ issue / wait counters (one pass of 8 SQ counters)
cd /tmp
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $O/pmc_attn -- python3 $R/tools/dattn_bench.py > $O/pmc_attn.log 2>&1
rm -f $O/pmc_attn/*/*kernel_trace.csv
cd $R
{ echo "# SQ counters of the pair-attention kernels (one pass of 8 SQ counters, tools/refresh_profiles.sh, summarised by tools/sq_summary.py):"
  echo "#   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -- python3 tools/dattn_bench.py"
  echo "# problems of tools/dattn_bench.py: (B 32, N 3136, H 4, hd 16) = DSEB-56x56 of the ACDC preset, (32, 784, 4, 32), (24, 3136, 8, 8), (24, 784, 8, 16)"
  python3 tools/sq_summary.py $(ls $O/pmc_attn/*/*counter_collection.csv | head -1); } > $O/${TAG}_attn_sq_counters.txt 2>&1
cp $O/prof_summary_c.txt $O/${TAG}_c_prof_summary.txt
cp $O/${TAG}_c_kernel_stats_noroofline.csv profiles/ 2>/dev/null
python3 tools/traffic_per_step.py $TAG > $O/${TAG}_traffic_per_step.txt 2>&1 || true
ls -la $O
