#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (ROCm 7.2 default output of
`rocprofv3 --kernel-trace --stats`) into the plain-text per-kernel summary that is
committed under profiles/."""
import sqlite3
import sys


def main(db_path, out_path, title=""):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), "
        "max(workgroup_x) from kernels group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows) or 1
    with open(out_path, "w") as f:
        if title:
            f.write(f"# {title}\n")
        f.write("# source: rocprofv3 --kernel-trace --stats (rocpd sqlite, durations in ns)\n")
        f.write("calls,total_ns,avg_ns,min_ns,max_ns,pct,vgpr,agpr,sgpr,lds_bytes,grid_x,wg_x,kernel\n")
        for r in rows:
            f.write(f"{r[1]},{int(r[2])},{r[3]:.0f},{int(r[4])},{int(r[5])},{100.0 * r[2] / total:.2f},"
                    f"{r[6]},{r[7]},{r[8]},{r[9]},{r[10]},{r[11]},\"{r[0]}\"\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
