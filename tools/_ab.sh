#!/bin/bash
# same-box A/B of an environment switch: tools/_ab.sh VAR [value]  (bench with VAR unset, then VAR=value, twice each, alternating)
V=$1; VAL=${2:-1}
for i in 1 2; do
  unset $V; echo -n "unset: "; timeout 200 python bench.py --no-cpu-baseline --no-f32 --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  export $V=$VAL; echo -n "$V=$VAL: "; timeout 200 python bench.py --no-cpu-baseline --no-f32 --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
unset $V
