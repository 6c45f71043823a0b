#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/t19; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/t19 -o t -- python $GRAFT_REPO_ROOT/tools/bench_train.py 19 256 hip > /tmp/t19.log 2>&1
F=$(find /tmp/t19 -name "t_kernel_stats.csv" | head -1)
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$F")))[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f} %")
PY
grep "train step" /tmp/t19.log
