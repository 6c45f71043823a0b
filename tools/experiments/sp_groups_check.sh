#!/bin/bash
# two host-thread groups against one (device-resident random streams): repeatability (round 6)
run() { echo -n "boards=$1 games=$2 groups=$3: "; TG_SP_LANES=1 python tools/bench_selfplay.py $1 400 $2 $3 2>&1 | tail -1 | sed 's/.*-> //'; }
for rep in 1 2 3; do run 16 256 2; done
run 16 256 1
run 8 128 1; run 8 128 2
run 12 192 1; run 12 192 2
run 24 256 1; run 24 256 2
run 32 320 1; run 32 320 2
run 16 256 3
