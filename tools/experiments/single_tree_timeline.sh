#!/bin/bash
# kernel timeline of single-tree moves (tools/bench_single_tree.py under rocprofv3 --kernel-trace): busy time, gaps, per-kernel totals
cd /tmp && export TMPDIR=/tmp
S=${1:-9}
O=$GRAFT_REPO_ROOT/gpurun_out/st_tl$S; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o tr -- python $GRAFT_REPO_ROOT/tools/bench_single_tree.py $S 4 > $O/log.txt 2>&1
F=$(find $O -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/experiments/sp_timeline.py $F 90 > $O/timeline.txt 2>&1
find $O -name "*.csv" -delete
head -120 $O/timeline.txt
