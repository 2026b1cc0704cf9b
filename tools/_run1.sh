set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
python -m pytest tests/test_wgrad_group.py tests/test_wellcond.py tests/test_ham_oracle.py tests/test_caller_protocol.py -q -m gpu -x 2>&1 | tail -40 > gpurun_out/r3a/new_tests.log
python -m pytest tests/test_modules_parity.py tests/test_bf16_mode.py tests/test_model_parity.py tests/test_graph_replay.py tests/test_overlap.py -q -m gpu 2>&1 | tail -30 > gpurun_out/r3a/old_tests.log
python bench.py --no-f32 --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r3a/bench_group.json 2> gpurun_out/r3a/bench_group.err
CENET_WGRAD_GROUP=0 python bench.py --no-f32 --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > gpurun_out/r3a/bench_nogroup.json 2> gpurun_out/r3a/bench_nogroup.err
tail -5 gpurun_out/r3a/new_tests.log; tail -3 gpurun_out/r3a/old_tests.log
python - <<'PY'
import json
for f in ("bench_group","bench_nogroup"):
    try:
        d=json.loads(open(f"gpurun_out/r3a/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["config"].get("launch"), d["config"].get("launch_choice"))
        if "roofline" in d: print(json.dumps(d["roofline"])[:1500])
    except Exception as e: print(f, "ERR", e)
PY
