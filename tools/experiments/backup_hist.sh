#!/bin/bash
# distribution of backup_kernel launch times in a self-play shard (is the one-wave fallback ever taken?)
cd /tmp && export TMPDIR=/tmp
B=${1:-64}
rm -rf /tmp/bh; rocprofv3 --kernel-trace --output-format csv -d /tmp/bh -o t -- python $GRAFT_REPO_ROOT/tools/bench_selfplay.py $B 400 $((B*2)) 1 > /tmp/bh.log 2>&1
F=$(find /tmp/bh -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv, collections
rows=[r for r in csv.DictReader(open("$F")) if "backup_kernel" in r["Kernel_Name"] or "select_gumbel" in r["Kernel_Name"]]
for name in ("backup_kernel", "select_gumbel"):
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if name in r["Kernel_Name"]]
    d.sort(); n=len(d)
    print(name, "n", n, "p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f us, sum %.1f ms; >150us: %d launches = %.1f ms" % (d[n//10], d[n//2], d[n*9//10], d[n*99//100], d[-1], sum(d)/1e3, sum(1 for x in d if x>150), sum(x for x in d if x>150)/1e3))
PY
tail -1 /tmp/bh.log | sed 's/.*-> //'
