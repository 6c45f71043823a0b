#!/bin/bash
# sub-group / forward-cap sweep of a self-play shard at the current kernels (tools/bench_selfplay.py)
run() { echo -n "boards=$1 games=$2 SUB=$3 CAP=$4: "; TG_SP_SUBGROUPS=$3 TG_SP_FWD_CAP=$4 python tools/bench_selfplay.py $1 400 $2 1 2>&1 | tail -1 | sed 's/.*-> //'; }
echo -n "boards=16 default: "; python tools/bench_selfplay.py 16 400 128 1 2>&1 | tail -1 | sed 's/.*-> //'
for g in 2 3 4 5; do run 16 128 $g 0; done
run 16 128 3 224; run 16 128 4 224
echo -n "boards=64 default: "; python tools/bench_selfplay.py 64 400 256 1 2>&1 | tail -1 | sed 's/.*-> //'
for g in 2 3 4; do for c in 208 224 240; do run 64 256 $g $c; done; done
