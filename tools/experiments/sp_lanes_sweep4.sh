#!/bin/bash
# repeatability of lanes = 2 against one lock-step group (round 6)
export TG_DEBUG_KNOBS=1
run() { # boards games lanes hwq
  echo -n "boards=$1 games=$2 lanes=$3 HWQ=${4:-default}: "
  if [ -n "$4" ]; then export GPU_MAX_HW_QUEUES=$4; else unset GPU_MAX_HW_QUEUES; fi
  TG_SP_LANES=$3 python tools/bench_selfplay.py $1 400 $2 1 2>&1 | tail -1 | sed 's/.*-> //'
}
for rep in 1 2 3; do run 16 256 1 8; run 16 256 2 8; run 16 256 2 16; done
run 8 128 1 8; run 8 128 2 8
run 24 256 1 8; run 24 256 2 8
run 32 320 1 8; run 32 320 2 8
