#!/bin/bash
# Round-6 kernel-trace evidence at the final code (one call on the GPU box): rocprofv3 --kernel-trace --stats of the headline bench
# (2 048 trees), of the single-tree searches (9x9, 19x19; tools/bench_single_tree.py) and of a 64-board self-play shard.
# Summaries -> gpurun_out/r06_final/, copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_final
rm -rf $OUT; mkdir -p $OUT
trace() { name=$1; shift; rm -rf /tmp/lt; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt -o t -- "$@" > $OUT/$name.log 2>&1
  f=$(find /tmp/lt -name t_kernel_stats.csv | head -1); [ -n "$f" ] && cp $f $OUT/r06_final_${name}_kernel_stats.csv; tail -2 $OUT/$name.log | cut -c1-300; }
trace bench_trees2048 python $R/bench.py --steps 3 --warmup 1 --trees 2048 --no-cpu-baseline --no-legs
trace single_tree_9x9 python $R/tools/bench_single_tree.py 9 8
trace single_tree_19x19 python $R/tools/bench_single_tree.py 19 4
trace selfplay_64_boards python $R/tools/bench_selfplay.py 64 400 128 1
echo done
# (round 6) the 16-board shard - two host-thread groups by default - with its concurrency picture
trace selfplay_16_boards python $R/tools/bench_selfplay.py 16 400 96
rm -rf /tmp/lt2; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt2 -o t -- python $R/tools/bench_selfplay.py 16 400 64 > $OUT/sp16_trace.log 2>&1
f=$(find /tmp/lt2 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/experiments/sp_timeline.py $f 220 > $OUT/r06_selfplay_16_boards_two_groups_timeline.txt
echo done16
