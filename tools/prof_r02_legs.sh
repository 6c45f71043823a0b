#!/bin/bash
# Round-2 kernel-trace evidence for the bench legs (one call on the GPU box):  bash tools/prof_r02_legs.sh
# rocprofv3 --kernel-trace --stats of the single-tree searches (9x9 PUCT + Gumbel, 19x19 PUCT), of the 16- and
# 64-board self-play shards, and SQ / HBM counters of the 19x19 forward kernel.  Summaries -> gpurun_out/legs_r02/,
# to be copied into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/legs_r02
rm -rf $OUT; mkdir -p $OUT
trace() { name=$1; shift; rm -rf /tmp/lt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt -o t -- "$@" > $OUT/$name.log 2>&1
  f=$(find /tmp/lt -name t_kernel_stats.csv | head -1); [ -n "$f" ] && cp $f $OUT/r02_${name}_kernel_stats.csv; grep -E "selfplay boards|MCTSTree|19x19 search" $OUT/$name.log | tail -2; }
trace single_tree_9x9 python $R/tools/bench_api_latency.py
trace single_tree_19x19 python $R/tools/bench_api_latency_19.py
trace selfplay_16_boards python $R/tools/bench_selfplay.py 16 400 64 1
trace selfplay_64_boards python $R/tools/bench_selfplay.py 64 400 128 1
pass() { dir=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$dir -o p -- python $R/tools/bench_net.py 19 4096 > $OUT/$dir.log 2>&1; echo "$dir rc=$?"; }
pass f19_a SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
pass f19_c FETCH_SIZE
pass f19_d WRITE_SIZE
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
def rows(d):
    f = glob.glob(f"{out}/{d}/**/p_counter_collection.csv", recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
def kernel_rows(d):
    return [r for r in rows(d) if "dualnet_fwd_split_kernel" in r.get("Kernel_Name", "")]
def mean_counter(d, name):                       # per-launch average, as tools/pmc_r02_summary.py takes it
    v = [float(r["Counter_Value"]) for r in kernel_rows(d) if r["Counter_Name"] == name]
    return sum(v) / len(v) if v else None
res = {"kernel": "dualnet_fwd_split_kernel<19, 1, f16x2>", "positions_per_launch": 4096}
busy, gui = mean_counter("f19_a", "SQ_VALU_MFMA_BUSY_CYCLES"), mean_counter("f19_a", "GRBM_GUI_ACTIVE")
if busy and gui:
    res["mfma_busy_fraction_of_simd_cycles"] = busy / (256 * 4 * gui / 8)   # 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
fetch, write = mean_counter("f19_c", "FETCH_SIZE"), mean_counter("f19_d", "WRITE_SIZE")
if fetch is not None and write is not None:
    # same corrections as tools/pmc_r02_summary.py (MI355X_MICROARCH.md, HBM section): counters in KB, FETCH_SIZE doubled
    hbm = (2 * fetch + write) * 1024
    res["hbm_side_bytes_per_position"] = hbm / 4096
    res["algorithmic_io_bytes_per_position"] = 6 * 361 * 4 + 362 * 4 + 12
    res["note"] = ("two thirds of a board's residual image (93 KB, written 7x and read 6x per position) live in a "
                   "per-workgroup scratch in global memory (the rest in LDS); with the 1.9 MB weight stream they exceed the "
                   "4 MB L2 of an XCD (32 workgroups), so part of them travels to the Infinity Cache and back")
json.dump(res, open(f"{out}/r02_pmc_forward_split_19x19_b4096.json", "w"), indent=1)
print(res)
PY
