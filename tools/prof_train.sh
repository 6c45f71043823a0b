#!/bin/bash
# rocprofv3 kernel trace of the training step (eager + hipGraph legs of tools/bench_train.py)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_train
rm -rf $OUT; mkdir -p $OUT
export MIOPEN_FIND_MODE=2   # heuristic solver choice: keeps MIOpen's search benchmarks out of the trace
rocprofv3 --kernel-trace --stats -d $OUT -o tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py 9 256 > $OUT/tr.log 2>&1
tail -3 $OUT/tr.log | cut -c1-300
DB=$(find $OUT -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $OUT/summary.csv "bench_train.py 9 256 (RL step, batch 256: 71 steps: eager, hipGraph warm-up and replays; MIOPEN_FIND_MODE=2)" > $OUT/summary.txt 2>&1; head -30 $OUT/summary.txt
rm -f $DB
