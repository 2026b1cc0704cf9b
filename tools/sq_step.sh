R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/sqstep -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --graph off > $R/gpurun_out/sqstep.log 2>&1
cd $R
python3 - <<'P'
import csv,glob,collections
f=glob.glob('gpurun_out/sqstep/*/*counter_collection.csv')[0]
rows=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].replace("void ","").split("(")[0]
    rows[n][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_WAVE_CYCLES":
        rows[n]["_dur"]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"]); rows[n]["_n"]+=1
out=[]
for n,c in rows.items():
    wc=c["SQ_WAVE_CYCLES"]
    if not wc: continue
    out.append((c["_dur"],n,c))
out.sort(reverse=True)
with open('gpurun_out/sqstep_summary.txt','w') as fo:
    for dur,n,c in out[:70]:
        wc=c["SQ_WAVE_CYCLES"]
        fo.write(f"{dur/3/1e6:7.3f} ms/step {int(c['_n']/3):4d}x  wait {c['SQ_WAIT_ANY']/wc:.2f} stall {c['SQ_WAIT_INST_ANY']/wc:.2f} act {c['SQ_ACTIVE_INST_ANY']/wc:.2f} valu {c['SQ_ACTIVE_INST_VALU']/wc:.2f}  {n[:80]}\n")
P
rm -rf gpurun_out/sqstep
