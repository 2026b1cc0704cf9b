#!/bin/bash
# Sweeps the ring GEMM's tile shape / split multiplier over the GEMM shapes of one training step (tools/gemm_replay.py) and
# prints, per configuration, the isolated time of every distinct shape: one file per configuration under gpurun_out/sweep/.
mkdir -p gpurun_out/sweep
for t in auto 64x64 128x64 64x128 128x128; do
  for m in 1 0.5 2; do
    if [ "$t" = auto ]; then unset CENET_RING_TILE; else export CENET_RING_TILE=$t; fi
    export CENET_RING_SPLIT_MUL=$m
    python tools/gemm_replay.py 400 > gpurun_out/sweep/${t}_$m.txt 2>&1
    grep "isolated total" gpurun_out/sweep/${t}_$m.txt | sed "s/^/$t x$m: /"
  done
done
