#!/bin/bash
# one SQ pass with LDS counters over two eager steps; prints the kernels matching $1 (regex)   bash tools/sq_lds.sh 'dw3x3|dwact'
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R
PAT=${1:-dw3x3}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/sqlds -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --graph off > $R/gpurun_out/sqlds.log 2>&1
cd $R
python3 - "$PAT" <<'P'
import csv,glob,collections,re,sys
pat=re.compile(sys.argv[1])
fs=glob.glob('gpurun_out/sqlds/*/*counter_collection.csv')
if not fs: print(open('gpurun_out/sqlds.log').read()[-1500:]); sys.exit()
rows=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(fs[0])):
    n=r["Kernel_Name"].replace("void ","").split("(")[0]
    if not pat.search(n): continue
    key=(n,r["Grid_Size"])
    rows[key][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_WAVE_CYCLES":
        rows[key]["_dur"]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"]); rows[key]["_n"]+=1
for (n,g),c in sorted(rows.items(),key=lambda kv:-kv[1]["_dur"]):
    wc=c["SQ_WAVE_CYCLES"] or 1
    print(f"{c['_dur']/max(c['_n'],1)/1e3:7.1f} us x{int(c['_n']):3d} grid {g:>8s} wait {c['SQ_WAIT_ANY']/wc:.2f} ldswait {c['SQ_WAIT_INST_LDS']/wc:.2f} lds_act {c['SQ_ACTIVE_INST_LDS']/wc:.2f} valu_act {c['SQ_ACTIVE_INST_VALU']/wc:.2f} "
          f"bankconf/ldsinst {c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_INSTS_LDS'],1):.2f} lds/valu {c['SQ_INSTS_LDS']/max(c['SQ_INSTS_VALU'],1):.2f}  {n[:60]}")
P
rm -rf gpurun_out/sqlds
