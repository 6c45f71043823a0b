#!/bin/bash
# kernel trace of a 16-board shard with 1 and 2 lanes: tools/experiments/sp_timeline.py windows
cd /tmp && export TMPDIR=/tmp
export TG_DEBUG_KNOBS=1
R=$GRAFT_REPO_ROOT
for cfg in "1 4" "2 16"; do
  set -- $cfg
  OUT=$R/gpurun_out/trace_l$1; rm -rf $OUT; mkdir -p $OUT
  GPU_MAX_HW_QUEUES=$2 TG_SP_LANES=$1 rocprofv3 --kernel-trace --output-format csv -d $OUT -o sp -- python $R/tools/bench_selfplay.py ${BOARDS:-16} 400 ${GAMES:-48} 1 > $OUT/sp.log 2>&1
  tail -1 $OUT/sp.log | cut -c1-200
  CSV=$(find $OUT -name "*kernel_trace.csv" | head -1)
  python $R/tools/experiments/sp_timeline.py $CSV 260 > $R/gpurun_out/r06_sp_timeline_lanes$1.txt
  head -20 $R/gpurun_out/r06_sp_timeline_lanes$1.txt
  rm -rf $OUT
done
