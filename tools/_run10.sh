set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
(time python -m pytest tests -q -m gpu 2>&1 | tail -25) > $O/gpu_all.log 2>&1
tail -12 $O/gpu_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3i/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["launch"])
print(json.dumps(d["roofline"]["weight_gradient_kernels"])[:1800])
print(d["roofline_stages"].get("grouped_weight_gradients"), d["roofline_stages"]["patch_embed1"])
print(d.get("cpu_baseline"))
PY
