#!/bin/bash
# lanes x sub-groups x forward cap, device-resident random streams, GPU_MAX_HW_QUEUES=16 (round 6)
export TG_DEBUG_KNOBS=1
export GPU_MAX_HW_QUEUES=${HWQ:-16}
run() { # boards games lanes sub cap
  echo -n "boards=$1 games=$2 lanes=$3 sub=${4:-auto} cap=${5:-auto}: "
  if [ -n "$4" ]; then export TG_SP_SUBGROUPS=$4; else unset TG_SP_SUBGROUPS; fi
  if [ -n "$5" ]; then export TG_SP_FWD_CAP=$5; else unset TG_SP_FWD_CAP; fi
  TG_SP_LANES=$3 python tools/bench_selfplay.py $1 400 $2 1 2>&1 | tail -1 | sed 's/.*-> //'
}
run 16 192 2 1
run 16 192 2 3
run 16 192 3
run 16 192 4 1
run 16 192 4 2
run 64 512 2 2 112
run 64 512 2 2 96
run 64 512 2 1 112
run 64 512 4 1 56
run 64 512 4 2 56
run 64 512 4 3 0
