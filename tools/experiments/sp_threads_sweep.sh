#!/bin/bash
# host-thread groups / lanes with more hardware queues, device-resident random streams (round 6)
export TG_DEBUG_KNOBS=1
run() { echo -n "boards=$1 games=$2 groups=$3 lanes=$4 HWQ=${5:-default}: "
  if [ -n "$5" ]; then export GPU_MAX_HW_QUEUES=$5; else unset GPU_MAX_HW_QUEUES; fi
  TG_SP_LANES=$4 python tools/bench_selfplay.py $1 400 $2 $3 2>&1 | tail -1 | sed 's/.*-> //'; }
run 16 192 1 1
run 16 192 1 1 16
run 16 192 2 1 16
run 16 192 1 2 16
run 16 192 2 1 8
run 64 512 1 1 16
run 64 512 2 1 16
run 64 512 1 2 16
run 64 512 4 1 16
