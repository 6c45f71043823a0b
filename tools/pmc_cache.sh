#!/bin/bash
# L2 / L1 request counters of the fused forward kernel (separate --pmc passes, kernel-trace only).
# Writes gpurun_out/pmc_cache/summary.txt (per-launch averages over the launches of bench_net.py).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_cache
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/tools/bench_net.py 9 65536 > $OUT/$name.log 2>&1; }
run a TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
run b TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
run c TA_BUSY_avr TCC_BUSY_avr TCP_PENDING_STALL_CYCLES_sum
python3 - <<'PY' | tee $OUT/summary.txt
import csv, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_cache"
c = {}
for g in "abc":
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{out}/{g}/p_counter_collection.csv")):
        if "dualnet_fwd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"])); kn = r["Kernel_Name"]
    for k, v in agg.items():
        c[k] = sum(v) / len(v)
dur = []
for r in csv.DictReader(open(f"{out}/a/p_kernel_trace.csv")):
    if "dualnet_fwd" in r["Kernel_Name"]:
        dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
ms = sum(dur) / len(dur) / 1e6
print("kernel", kn.split("(")[0], "positions/launch 65536, avg launch %.3f ms (profiled)" % ms)
for k in sorted(c):
    print(f"{k:34s} {c[k]:.4g}")
cyc = c["GRBM_GUI_ACTIVE"] / 8
print("L2 read requests x 128 B = %.1f GB per launch = %.2f TB/s; per CU %.1f B/clk" % (
    c["TCC_READ_sum"] * 128 / 1e9, c["TCC_READ_sum"] * 128 / ms / 1e9, c["TCC_READ_sum"] * 128 / 256 / cyc))
print("L2 hit rate %.4f; L1 (TCP) hit rate %.3f; TCC busy %.2f, TA busy %.2f of kernel cycles" % (
    c["TCC_HIT_sum"] / c["TCC_REQ_sum"], 1 - c["TCP_TCC_READ_REQ_sum"] * 2 / c["TCP_TOTAL_CACHE_ACCESSES_sum"],
    c["TCC_BUSY_avr"] / cyc, c["TA_BUSY_avr"] / cyc))
PY
