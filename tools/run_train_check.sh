#!/bin/bash
# training step: parity tests, timing, kernel stats (hip only)
O=gpurun_out/${1:-r05u}; mkdir -p $O
timeout 900 python -m pytest tests/test_train_step.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_train.txt
cat $O/pytest_train.txt | tail -5
timeout 300 python tools/bench_train.py 9 256,1024 2>&1 | tail -3 > $O/bench_train.txt; cat $O/bench_train.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py 9 256 hip > $GRAFT_REPO_ROOT/$O/tr.log 2>&1
DB=$(find $GRAFT_REPO_ROOT/$O/prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $GRAFT_REPO_ROOT/$O/summary.csv "bench_train.py 9 256 hip" > $GRAFT_REPO_ROOT/$O/summary.txt 2>&1; head -16 $GRAFT_REPO_ROOT/$O/summary.txt | cut -c1-150
rm -rf $GRAFT_REPO_ROOT/$O/prof
