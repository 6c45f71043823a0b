#!/bin/bash
# PMC passes for the fused forward kernel (separate runs per counter group, kernel-trace only).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_r1
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_VALU_MFMA_BUSY_CYCLES|SQ_BUSY_CYCLES|SQ_BUSY_CU_CYCLES|GRBM_GUI_ACTIVE|SQ_WAVE_CYCLES|SQ_INSTS_VALU_MFMA_MOPS_F32|SQ_INSTS_MFMA|SQ_INSTS_VALU_MFMA_F32|SQ_WAIT_INST_ANY|SQ_WAIT_ANY|SQ_ACTIVE_INST_ANY|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|FETCH_SIZE|WRITE_SIZE|SQ_INST_CYCLES_VMEM|SQ_ACTIVE_INST_LDS|SQ_INSTS_LDS|SQ_WAVES|MfmaUtil|SQ_INSTS_VALU)\b" | sort -u > $OUT/available.txt
cat $OUT/available.txt | tr '\n' ' '; echo
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/tools/bench_net.py 9 65536 > $OUT/$name.log 2>&1; tail -1 $OUT/$name.log; }
run a SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run b SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run c FETCH_SIZE
run d WRITE_SIZE
find $OUT -name "*.csv" | head -20
