#!/bin/bash
# average duration of the Gumbel selection launch in a self-play shard of 1 / 16 boards (rocprofv3 --stats) and the shard's rate
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in 1 16; do
  rm -rf /tmp/ab; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab -o t -- python $R/tools/bench_selfplay.py $b 400 $((b*4)) > /tmp/ab.log 2>&1
  f=$(find /tmp/ab -name t_kernel_stats.csv | head -1)
  echo "boards $b: $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'select_gumbel' in r['Name']: print(r['Calls'], 'launches, average', round(float(r['AverageNs'])/1e3,1), 'us'); break
")"
done
