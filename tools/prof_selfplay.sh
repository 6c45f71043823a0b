#!/bin/bash
# rocprofv3 kernel trace of the Gumbel self-play shard; summary via tools/rocpd_summary.py
cd /tmp && export TMPDIR=/tmp
BOARDS=${1:-256}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_sp$BOARDS
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o sp -- python $GRAFT_REPO_ROOT/tools/bench_selfplay.py $BOARDS 400 $((BOARDS*2)) > $OUT/sp.log 2>&1
tail -1 $OUT/sp.log | cut -c1-300
DB=$(find $OUT -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $OUT/summary.csv "bench_selfplay.py $BOARDS boards 400 visits" | head -16
