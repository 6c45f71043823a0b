#!/usr/bin/env python3
"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV: python tools/kernel_timeline.py t_kernel_trace.csv [rows]
Prints a window from the middle of the trace (gap before, duration, name) and the totals."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
show = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lo = len(rows) // 2
prev_end = None
for r in rows[lo:lo + show]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{gap:8.1f} us gap  {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:70]}")
    prev_end = max(e, prev_end or 0)
busy = gap_tot = 0
prev_end = None
hist = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None and s > prev_end:
        g = s - prev_end
        gap_tot += g
        b = "<10us" if g < 10e3 else "<30us" if g < 30e3 else "<100us" if g < 100e3 else ">=100us"
        hist[b] = hist.get(b, (0, 0))
        hist[b] = (hist[b][0] + 1, hist[b][1] + g)
    busy += (e - s) if prev_end is None or s >= prev_end else max(0, e - prev_end)
    prev_end = max(e, prev_end or 0)
print(f"kernels {len(rows)}  busy {busy / 1e6:.2f} ms  gaps {gap_tot / 1e6:.2f} ms")
for b, (n, t) in sorted(hist.items()):
    print(f"  gaps {b:8s}: {n:6d}  {t / 1e6:8.2f} ms")
