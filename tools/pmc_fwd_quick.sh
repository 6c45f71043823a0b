#!/bin/bash
# Quick counter comparison of forward kernels on the GPU box:  bash tools/pmc_fwd_quick.sh "s32 split16"
# (--kernel-trace + --pmc passes only).  Output: gpurun_out/pmc_quick/<algo>_<pass>/..., summary printed.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_quick
rm -rf $OUT; mkdir -p $OUT
for algo in $1; do
  export TG_FWD_ALGO=$algo
  pass() { dir=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${algo}_$dir -o p -- python $R/tools/bench_net.py 9 65536 > $OUT/${algo}_$dir.log 2>&1; echo "$algo $dir rc=$?"; }
  pass a SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
  pass b SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  pass f TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_BUSY_avr TCC_REQ_sum
done
python3 - <<PY
import csv, glob, collections, os
out="$OUT"
for algo in "$1".split():
    c=collections.defaultdict(list); dur=[]
    for d in ("a","b","f"):
        for path in glob.glob(f"{out}/{algo}_{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                if "dualnet_fwd" in r["Kernel_Name"] and "wino" not in r["Kernel_Name"]:
                    c[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for path in glob.glob(f"{out}/{algo}_{d}/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                if "dualnet_fwd" in r["Kernel_Name"] and "wino" not in r["Kernel_Name"]:
                    dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    m={k:sum(v)/len(v) for k,v in c.items()}
    dns=sum(dur)/len(dur)
    simd=256*4*m["GRBM_GUI_ACTIVE"]/8
    print(f"{algo}: {dns/1e6:.3f} ms profiled, clock {m['GRBM_GUI_ACTIVE']/8/dns:.3f} GHz, mfma_busy {m['SQ_VALU_MFMA_BUSY_CYCLES']/simd:.3f}, "
          f"wait_any {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.3f}, wait_inst {m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.3f}, active_inst {m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']:.3f}, "
          f"lds_conflict {m['SQ_LDS_BANK_CONFLICT']/max(1,m['SQ_LDS_IDX_ACTIVE']):.3f}, lds_active/cu_cycle {m['SQ_LDS_IDX_ACTIVE']/(256*m['GRBM_GUI_ACTIVE']/8):.3f}, "
          f"tcc_busy {m.get('TCC_BUSY_avr',0)/(m['GRBM_GUI_ACTIVE']/8):.3f}, l2_req_KB_per_pos {m.get('TCC_REQ_sum',0)*128/65536/1024:.1f}, tcp_tcc_read_KB_per_pos {m.get('TCP_TCC_READ_REQ_sum',0)*64/65536/1024:.1f}")
PY
