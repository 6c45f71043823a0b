#!/bin/bash
# lanes x sub-groups sweep of a self-play shard (round 6; tools/bench_selfplay.py: boards visits games groups)
export TG_DEBUG_KNOBS=1
run() { # boards games lanes sub
  echo -n "boards=$1 games=$2 LANES=$3 SUB=${4:-auto}: "
  if [ -n "$4" ]; then export TG_SP_SUBGROUPS=$4; else unset TG_SP_SUBGROUPS; fi
  TG_SP_LANES=$3 python tools/bench_selfplay.py $1 400 $2 1 2>&1 | tail -1 | sed 's/.*-> //'
}
for l in 1 2 4 8; do run 16 192 $l; done
for l in 2 4 8; do run 16 192 $l 1; done
for l in 1 2 4 8; do run 64 512 $l; done
for l in 2 4 8 16; do run 64 512 $l 1; done
