#!/bin/bash
export TG_DEBUG_KNOBS=1
export GPU_MAX_HW_QUEUES=16
run() { echo -n "boards=$1 games=$2 lanes=$3 HWQ=16: "; TG_SP_LANES=$3 python tools/bench_selfplay.py $1 400 $2 1 2>&1 | tail -1 | sed 's/.*-> //'; }
for b in "8 128" "12 192" "20 256" "24 256" "28 256" "32 320" "48 384"; do set -- $b; run $1 $2 1; run $1 $2 2; done
