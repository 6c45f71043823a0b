#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in warm2 nowarm; do
  OUT=$R/gpurun_out/trace_$m; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT -o sp -- python $R/tools/experiments/sp_pool_order.py $m > $OUT/sp.log 2>&1
  tail -1 $OUT/sp.log | cut -c1-200
  CSV=$(find $OUT -name "*kernel_trace.csv" | head -1)
  python $R/tools/experiments/sp_timeline.py $CSV 200 > $R/gpurun_out/r06_sp_timeline_$m.txt
  head -16 $R/gpurun_out/r06_sp_timeline_$m.txt
  python - <<PY
import csv,collections
rows=list(csv.DictReader(open("$CSV")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
main=rows[len(rows)//3:]
q=collections.Counter(r.get("Queue_Id") for r in main)
print("queues used in main part:", dict(q))
byq=collections.defaultdict(collections.Counter)
for r in main: byq[r.get("Queue_Id")][r["Kernel_Name"].split("(")[0][-40:]]+=1
for k,v in byq.items(): print(k, v.most_common(4))
PY
  rm -rf $OUT
done
