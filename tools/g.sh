#!/bin/bash
# build the product library, then run a command on the MI355X box: tools/g.sh <timeout-seconds> '<command>'
set -e
cd /root/repo
python -m cenet_amd.build > /tmp/build.log 2>&1 || { tail -30 /tmp/build.log; exit 1; }
T=$1; shift
timeout $((T + 1500)) /usr/local/graft/bin/gpurun --timeout $T -- "$@"
