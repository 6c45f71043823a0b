#!/usr/bin/env python3
"""Concurrency picture of a self-play kernel trace (rocprofv3 --kernel-trace CSV): device busy (union over queues), idle, per-kernel
totals, and a window of the trace with the queue each kernel ran on.   python tools/experiments/sp_timeline.py sp_kernel_trace.csv [rows]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
show = int(sys.argv[2]) if len(sys.argv) > 2 else 80
def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:44]
# main run = the second half of the trace (the first is warm-up)
t_lo = int(rows[len(rows) // 3]["Start_Timestamp"])
main = [r for r in rows if int(r["Start_Timestamp"]) >= t_lo]
span = int(main[-1]["End_Timestamp"]) - int(main[0]["Start_Timestamp"])
busy, end = 0, None
for r in main:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if end is None or s >= end:
        busy += e - s; end = e
    elif e > end:
        busy += e - end; end = e
tot = collections.defaultdict(lambda: [0, 0])
for r in main:
    k = short(r["Kernel_Name"]); tot[k][0] += 1; tot[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"main part: span {span/1e6:.1f} ms, device busy (union) {busy/1e6:.1f} ms = {100*busy/span:.1f} %, kernels {len(main)}")
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t/1e6:9.2f} ms = {100*t/span:5.1f} % of span  {n:7d} x {t/n/1e3:8.1f} us  {k}")
qs = {}
lo = len(rows) * 2 // 3
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:lo + show]:
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    try:
        wgs = int(r.get("Grid_Size", r.get("Grid_Size_X", 0))) // max(1, int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1))))
    except Exception:
        wgs = -1
    print(f"{(s - t0)/1e3:9.1f} us  +{(e - s)/1e3:7.1f} us  q{q:<2d} wg{wgs:<5d} {'  ' * q}{short(r['Kernel_Name'])}")
