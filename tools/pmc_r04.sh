#!/bin/bash
# Round-4 counter evidence, one call on the GPU box:  bash tools/pmc_r04.sh
#   1. forward kernels (tools/bench_net.py 9 65536; default = w1d, then wsplit / split16 / wino): SQ / LDS / L2 / HBM counters in separate --pmc passes
#   2. tree kernels (bench.py --trees 2048, 3 steps): HBM-side and L2 bytes of select / backup / root / play
#   3. stand-alone featurise kernel (tools/bench_featurize.py)
#   4. kernel trace of the headline bench (--stats)
# Every pass is --kernel-trace + --pmc only (no other trace domain).  Summaries -> gpurun_out/pmc_r04/*.json|csv,
# to be copied into profiles/ (tools/pmc_r04_summary.py does the arithmetic and stamps the csrc digest).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_r04
rm -rf $OUT; mkdir -p $OUT
pass() { dir=$1; shift; cmd=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$dir -o p -- $cmd > $OUT/$dir.log 2>&1; echo "$dir: rc=$? $(tail -1 $OUT/$dir.log | cut -c1-160)"; }
# PMC_ONLY=tree: only the tree / featurise passes and the kernel trace (forward sources unchanged: their summaries stay valid)
FWD="python $R/tools/bench_net.py 9 65536"
if [ "$PMC_ONLY" != "tree" ]; then
pass fwd_a "$FWD" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass fwd_b "$FWD" SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass fwd_g "$FWD" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU
pass fwd_c "$FWD" FETCH_SIZE
pass fwd_d "$FWD" WRITE_SIZE
pass fwd_e "$FWD" TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
pass fwd_f "$FWD" TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_BUSY_avr
# the 2-D Winograd split kernel (TG_FWD_ALGO=wsplit: the 9x9 default of the first half of round 4) on the same batch
export TG_FWD_ALGO=wsplit
pass ws_a "$FWD" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass ws_b "$FWD" SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass ws_g "$FWD" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU
pass ws_c "$FWD" FETCH_SIZE
pass ws_d "$FWD" WRITE_SIZE
pass ws_e "$FWD" TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
pass ws_f "$FWD" TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_BUSY_avr
unset TG_FWD_ALGO
# the direct split kernel (TG_FWD_ALGO=split16, the 9x9 default up to round 3) on the same batch
export TG_FWD_ALGO=split16
pass w2_a "$FWD" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass w2_b "$FWD" SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass w2_c "$FWD" FETCH_SIZE
pass w2_d "$FWD" WRITE_SIZE
pass w2_e "$FWD" TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
pass w2_f "$FWD" TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_BUSY_avr
unset TG_FWD_ALGO
fi
# the exact-fp32 Winograd kernel (TG_FWD_ALGO=wino: bench.py's fp32_exact leg) on the same batch
if [ "$PMC_ONLY" != "tree" ] || [ "$PMC_WINO" = "1" ]; then
export TG_FWD_ALGO=wino
pass wn_a "$FWD" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass wn_b "$FWD" SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass wn_c "$FWD" FETCH_SIZE
pass wn_d "$FWD" WRITE_SIZE
pass wn_e "$FWD" TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
pass wn_f "$FWD" TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_BUSY_avr
unset TG_FWD_ALGO
fi
if [ "$PMC_ONLY" != "tree" ]; then
# 19x19 split kernel
F19="python $R/tools/bench_net.py 19 4096"
pass f19_a "$F19" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass f19_c "$F19" FETCH_SIZE
pass f19_d "$F19" WRITE_SIZE
pass f19_e "$F19" TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
fi
if [ "$PMC_ONLY" != "tree" ]; then
# banded 19x19 kernel: one tree's mini-batch (64 positions over 256 workgroups)
B19="python $R/tools/bench_net.py 19 64"
pass b19_a "$B19" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass b19_c "$B19" FETCH_SIZE
pass b19_d "$B19" WRITE_SIZE
pass b19_e "$B19" TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
fi
TREE="python $R/bench.py --steps 2 --warmup 1 --trees 2048 --no-cpu-baseline --no-legs"
pass tree_c "$TREE" FETCH_SIZE
pass tree_d "$TREE" WRITE_SIZE
pass tree_e "$TREE" TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum
FEAT="python $R/tools/bench_featurize.py"
pass feat_c "$FEAT" FETCH_SIZE
pass feat_d "$FEAT" WRITE_SIZE
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --trees 2048 --no-cpu-baseline --no-legs > $OUT/trace.log 2>&1
echo "trace: rc=$?"
# kernel traces of the single-tree legs (9x9 and 19x19 through the API)
for leg in single_tree_9x9:bench_api_latency.py single_tree_19x19:bench_api_latency_19.py; do
  name=${leg%%:*}; script=${leg##*:}; rm -rf /tmp/lt
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt -o t -- python $R/tools/$script > $OUT/$name.log 2>&1
  f=$(find /tmp/lt -name t_kernel_stats.csv | head -1); [ -n "$f" ] && cp $f $OUT/r04_${name}_kernel_stats.csv; grep -E "MCTSTree|19x19 search|per move" $OUT/$name.log | tail -1
done
python3 $R/tools/pmc_r04_summary.py $OUT
