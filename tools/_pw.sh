R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R
python3 $R/tools/probe_convfwd.py
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pw -- python3 $R/tools/probe_convfwd.py > $R/gpurun_out/pw.log 2>&1
cd $R; python3 tools/sq_summary.py $(ls gpurun_out/pw/*/*counter_collection.csv | head -1)
rm -rf gpurun_out/pw
