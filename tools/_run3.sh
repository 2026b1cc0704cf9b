set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
python -m pytest tests -q -m gpu -x 2>&1 | tail -40 > $O/gpu_tests.log
tail -15 $O/gpu_tests.log
python bench.py --no-f32 --no-cpu-baseline --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3c/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("launch"), d["config"].get("launch_choice"))
for k in d["roofline"]["top_kernels"]: print({a:b for a,b in k.items() if a not in ("traffic_source",)})
print(d.get("whole_step"))
PY
echo "== tile class sweep"
python tools/wgrad_bench.py 10 2>/dev/null
