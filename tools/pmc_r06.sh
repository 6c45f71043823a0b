#!/bin/bash
# Round-6 counter evidence, one call on the GPU box:  bash tools/pmc_r06.sh     (PMC_ONLY=f19 | fwd | tree restricts the passes)
#   1. 9x9 forward (tools/bench_net.py 9 65536; default = w1d, then wino): SQ / LDS / L2 / HBM counters in separate --pmc passes
#   2. 19x19 forward: the pair kernel on 4 096 boards and on one tree's mini-batch (64)
#   3. tree kernels (bench.py --trees 2048), stand-alone featurise kernel
#   4. kernel traces (--stats): headline bench, single-tree legs
# Every pass is --kernel-trace + --pmc only (no other trace domain).  Summaries -> gpurun_out/pmc_r06/*.json|csv, to be copied
# into profiles/ (tools/pmc_r06_summary.py does the arithmetic and stamps the csrc digest).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_r06
mkdir -p $OUT
pass() { dir=$1; shift; cmd=$1; shift; rm -rf $OUT/$dir; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$dir -o p -- $cmd > $OUT/$dir.log 2>&1; echo "$dir: rc=$? $(tail -1 $OUT/$dir.log | cut -c1-160)"; }
fwd_passes() {  # prefix, command
  pass $1_a "$2" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
  pass $1_b "$2" SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  pass $1_g "$2" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU
  pass $1_c "$2" FETCH_SIZE
  pass $1_d "$2" WRITE_SIZE
  pass $1_e "$2" TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
  pass $1_f "$2" TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_BUSY_avr
}
want() { [ -z "$PMC_ONLY" ] || [[ ",$PMC_ONLY," == *",$1,"* ]]; }
if want fwd; then
  fwd_passes fwd "python $R/tools/bench_net.py 9 65536"
  export TG_FWD_ALGO=wino
  fwd_passes wn "python $R/tools/bench_net.py 9 65536"
  unset TG_FWD_ALGO
fi
if want f19; then
  fwd_passes f19 "python $R/tools/bench_net.py 19 4096"
  fwd_passes b19 "python $R/tools/bench_net.py 19 64"
fi
if want tree; then
  TREE="python $R/bench.py --steps 2 --warmup 1 --trees 2048 --no-cpu-baseline --no-legs"
  pass tree_c "$TREE" FETCH_SIZE
  pass tree_d "$TREE" WRITE_SIZE
  pass tree_e "$TREE" TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum
  FEAT="python $R/tools/bench_featurize.py"
  pass feat_c "$FEAT" FETCH_SIZE
  pass feat_d "$FEAT" WRITE_SIZE
fi
if want trace; then
  rm -rf $OUT/trace
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --trees 2048 --no-cpu-baseline --no-legs > $OUT/trace.log 2>&1
  echo "trace: rc=$?"
  for leg in single_tree_9x9:bench_api_latency.py single_tree_19x19:bench_api_latency_19.py; do
    name=${leg%%:*}; script=${leg##*:}; rm -rf /tmp/lt
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt -o t -- python $R/tools/$script > $OUT/$name.log 2>&1
    f=$(find /tmp/lt -name t_kernel_stats.csv | head -1); [ -n "$f" ] && cp $f $OUT/r06_${name}_kernel_stats.csv; grep -E "MCTSTree|19x19 search|per move" $OUT/$name.log | tail -1
  done
fi
python3 $R/tools/pmc_r06_summary.py $OUT
