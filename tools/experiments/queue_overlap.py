#!/usr/bin/env python3
"""Do kernels of one hardware queue ever overlap in a rocprofv3 kernel trace?  python tools/experiments/queue_overlap.py trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
byq = collections.defaultdict(list)
for r in rows:
    byq[(r.get("Queue_Id"), r.get("Stream_Id", "-"))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
for q, ks in sorted(byq.items()):
    ks.sort()
    over = 0; worst = 0; ex = None
    for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
        if s1 < e0:
            over += 1
            if e0 - s1 > worst: worst = e0 - s1; ex = (n0, n1)
    print(f"queue/stream {q}: {len(ks)} kernels, {over} start before their predecessor ended (worst overlap {worst / 1e3:.1f} us {ex})")
