#!/bin/bash
export TG_DEBUG_KNOBS=1
run() { # boards games lanes sub cap hwq
  echo -n "boards=$1 games=$2 lanes=$3 sub=${4:-auto} cap=${5:-auto} HWQ=$6: "
  if [ -n "$4" ] && [ "$4" != "-" ]; then export TG_SP_SUBGROUPS=$4; else unset TG_SP_SUBGROUPS; fi
  if [ -n "$5" ] && [ "$5" != "-" ]; then export TG_SP_FWD_CAP=$5; else unset TG_SP_FWD_CAP; fi
  export GPU_MAX_HW_QUEUES=$6
  TG_SP_LANES=$3 python tools/bench_selfplay.py $1 400 $2 1 2>&1 | tail -1 | sed 's/.*-> //'
}
run 16 256 2 - - 32
run 16 256 2 3 - 32
run 16 256 4 1 - 32
run 16 256 4 2 - 32
run 16 256 3 - - 32
run 64 512 2 - - 32
run 64 512 2 2 96 32
run 64 512 4 - - 32
run 64 512 4 1 56 32
