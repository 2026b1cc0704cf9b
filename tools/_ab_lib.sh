#!/bin/bash
# same-box A/B of two builds of the library: ab_old.so / ab_new.so at the repo root (built from two source states)
for i in 1 2 3; do
  for v in old new; do
    cp ab_$v.so cenet_amd/libcenet_hip.so
    echo -n "$v: "; timeout 200 python bench.py --no-cpu-baseline --no-f32 --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
