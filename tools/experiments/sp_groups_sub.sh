#!/bin/bash
# sub-groups inside each of two host-thread groups (round 6)
export TG_DEBUG_KNOBS=1
run() { echo -n "boards=$1 games=$2 groups=2 sub=$3: "; TG_SP_SUBGROUPS=$3 python tools/bench_selfplay.py $1 400 $2 2 2>&1 | tail -1 | sed 's/.*-> //'; }
for s in 1 2 3; do run 16 256 $s; done
for s in 1 2 3; do run 24 256 $s; done
