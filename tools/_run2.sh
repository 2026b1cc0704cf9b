set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
for t in 64x64x4 128x128x2 128x128x3 128x64x3 64x128x3; do
  echo "== tile $t" >> $O/wgrad_bench.log
  CENET_GROUP_TILE=$t python tools/wgrad_bench.py 10 2>/dev/null >> $O/wgrad_bench.log
done
for it in 768 3072; do
  echo "== tile 64x64x4 items $it" >> $O/wgrad_bench.log
  CENET_GROUP_ITEMS=$it python tools/wgrad_bench.py 10 2>/dev/null >> $O/wgrad_bench.log
  echo "== tile 128x128x2 items $it" >> $O/wgrad_bench.log
  CENET_GROUP_TILE=128x128x2 CENET_GROUP_ITEMS=$it python tools/wgrad_bench.py 10 2>/dev/null >> $O/wgrad_bench.log
done
cat $O/wgrad_bench.log
python - <<'PY' 2>&1 | grep -v "^No \|amdgpu.ids" > gpurun_out/r3b/wellcond_diag.log
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import torch
import test_wellcond as T
from backend import use_hip
z=T.golden(); dev=use_hip()
for bf in (False, True, True):
    loss, lt, grads, flat, bufs = T._train_step(z, dev, bf)
    print("bf16" if bf else "fp32", loss, float(z["loss64"]))
    for seg, r in T.compare(z, grads).items(): print("   ", seg, r)
    if bf:
        if 'prev' in globals(): print("   run-to-run cos", torch.nn.functional.cosine_similarity(prev.double(), flat.double(), dim=0).item(), (prev-flat).abs().max().item())
        prev = flat
print("ref own", T.reference_fp32_error(z))
PY
cat gpurun_out/r3b/wellcond_diag.log
cd /tmp && export TMPDIR=/tmp
WGRAD_SETS=stage1,stage3 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- python3 $GRAFT_REPO_ROOT/tools/wgrad_bench.py 3 > $GRAFT_REPO_ROOT/$O/pmc_fetch.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/r3b/pmc_fetch/*/*counter_collection.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name']=='FETCH_SIZE': agg[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'gemm_group' in k: print(k, len(v), [round(x*2/1024,1) for x in v[:12]], 'MB (2xFETCH)')
PY
rm -rf $O/pmc_fetch
