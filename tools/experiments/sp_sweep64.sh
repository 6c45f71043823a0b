#!/bin/bash
export TG_DEBUG_KNOBS=1
run() { echo -n "boards=$1 games=$2 SUB=$3 CAP=$4: "; TG_SP_SUBGROUPS=$3 TG_SP_FWD_CAP=$4 python tools/bench_selfplay.py $1 400 $2 1 2>&1 | tail -1 | sed 's/.*-> //'; }
echo -n "boards=64 default: "; python tools/bench_selfplay.py 64 400 512 1 2>&1 | tail -1 | sed 's/.*-> //'
for g in 2 3 4; do for c in 216 224 232; do run 64 512 $g $c; done; done
run 64 512 2 0
run 64 512 1 0
