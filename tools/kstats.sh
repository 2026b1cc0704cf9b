#!/bin/bash
# per-kernel average durations of the benched step (rocprofv3 kernel trace of a short bench run): bash tools/kstats.sh <tag>
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/kstats_$TAG
mkdir -p $O
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32 --no-roofline > $O/st.log 2>&1
cp $(ls $O/st/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/st
tail -1 $O/st.log | cut -c1-200
