#!/bin/bash
# rocprofv3 kernel trace of the headline bench; summary goes to profiles/ via tools/rocpd_summary.py
cd /tmp && export TMPDIR=/tmp
TREES=${1:-256}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_t$TREES
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --trees $TREES --no-cpu-baseline --no-legs > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-400
ls $OUT
