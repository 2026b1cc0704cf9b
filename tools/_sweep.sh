#!/bin/bash
# same-box sweep of environment knobs: each line "VAR=value"; baseline (nothing set) first and last
run() { timeout 200 python bench.py --no-cpu-baseline --no-f32 --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo -n "baseline: "; run
for kv in "$@"; do
  echo -n "$kv: "; env $kv bash -c "$(declare -f run); run"
done
echo -n "baseline: "; run
