#!/bin/bash
# per-shape tile sweep of the ring GEMM (round 6): tools/gemm_replay.py under every forced tile, one file per tile in gpurun_out/sweep6/
mkdir -p gpurun_out/sweep6
for t in auto 64x64 128x64 64x128 128x128; do
  if [ "$t" = auto ]; then unset CENET_RING_TILE; else export CENET_RING_TILE=$t; fi
  timeout 300 python tools/gemm_replay.py 400 > gpurun_out/sweep6/$t.txt 2>&1
  grep "isolated total" gpurun_out/sweep6/$t.txt | sed "s/^/$t: /"
done
