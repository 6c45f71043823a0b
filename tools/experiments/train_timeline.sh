#!/bin/bash
# kernel trace of the training step with the side stream: which launches overlap (tools/experiments/sp_timeline.py)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/train_tl; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py 9 256 hip > $O/log.txt 2>&1
F=$(find $O -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/experiments/sp_timeline.py $F 70 > $O/timeline.txt 2>&1
rm -f $F; find $O -name "*.csv" -delete
head -90 $O/timeline.txt
