#!/bin/bash
export TG_DEBUG_KNOBS=1
run() { # boards games lanes queues
  echo -n "boards=$1 games=$2 LANES=$3 HWQ=${4:-default}: "
  if [ -n "$4" ]; then export GPU_MAX_HW_QUEUES=$4; else unset GPU_MAX_HW_QUEUES; fi
  TG_SP_LANES=$3 python tools/bench_selfplay.py $1 400 $2 1 2>&1 | tail -1 | sed 's/.*-> //'
}
echo "--- timing, 16 boards, 1 lane"; TG_SP_TIMING=1 TG_SP_LANES=1 python tools/bench_selfplay.py 16 400 64 1 2>&1 | grep -i "timing" | tail -3
echo "--- timing, 16 boards, 4 lanes"; TG_SP_TIMING=1 TG_SP_LANES=4 python tools/bench_selfplay.py 16 400 64 1 2>&1 | grep -i "timing" | tail -6
for q in 8 16 24; do for l in 1 2 4; do run 16 192 $l $q; done; done
for q in 8 16; do for l in 1 2 4; do run 64 512 $l $q; done; done
