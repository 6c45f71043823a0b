#!/bin/bash
# Sub-group count x forward cap sweep of a self-play shard (tools/bench_selfplay.py): leaf evaluations/s.
# usage: sp_subgroup_sweep.sh BOARDS VISITS GAMES
export TG_DEBUG_KNOBS=1
B=${1:-16}; V=${2:-400}; N=${3:-64}
echo "default: $(python tools/bench_selfplay.py $B $V $N 2>&1 | grep -o '[0-9]* leaf-evals/s')"
for g in 1 2 3 4 5 6; do
  for cap in 0 160 200 224 240; do
    r=$(TG_SP_SUBGROUPS=$g TG_SP_FWD_CAP=$cap python tools/bench_selfplay.py $B $V $N 2>&1 | grep -o '[0-9]* leaf-evals/s')
    echo "sub-groups $g cap $cap: $r"
  done
done
