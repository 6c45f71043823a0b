#!/bin/bash
# Gumbel selection workers per tree (TG_GUMBEL_WORKERS) on the self-play shard (tools/bench_selfplay.py)
export TG_DEBUG_KNOBS=1
for b in "16 192" "64 512"; do set -- $b
  for w in 6 10 15; do
    echo -n "boards=$1 workers=$w: "; TG_GUMBEL_WORKERS=$w python tools/bench_selfplay.py $1 400 $2 1 2>&1 | tail -1 | sed 's/.*-> //'
  done
done
