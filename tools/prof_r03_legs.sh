#!/bin/bash
# Round-3 kernel-trace evidence for the bench legs (one call on the GPU box):  bash tools/prof_r03_legs.sh
# rocprofv3 --kernel-trace --stats of the single-tree searches (9x9 PUCT + Gumbel, 19x19 PUCT) and of the 16- and
# 64-board self-play shards; forward accuracy and phase timelines.  Summaries -> gpurun_out/legs_r03/, to be copied
# into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/legs_r03
rm -rf $OUT; mkdir -p $OUT
trace() { name=$1; shift; rm -rf /tmp/lt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt -o t -- "$@" > $OUT/$name.log 2>&1
  f=$(find /tmp/lt -name t_kernel_stats.csv | head -1); [ -n "$f" ] && cp $f $OUT/r03_${name}_kernel_stats.csv; grep -E "selfplay boards|MCTSTree|19x19 search" $OUT/$name.log | tail -2; }
trace single_tree_9x9 python $R/tools/bench_api_latency.py
trace single_tree_19x19 python $R/tools/bench_api_latency_19.py
trace selfplay_16_boards python $R/tools/bench_selfplay.py 16 400 64 1
trace selfplay_64_boards python $R/tools/bench_selfplay.py 64 400 128 1
cd $R
python tools/check_forward_accuracy.py > $OUT/r03_forward_accuracy.txt 2>&1
(for a in split16 w2; do TG_FWD_ALGO=$a python tools/phase_profile.py 65536; done; TG_FWD_ALGO=split16 python tools/phase_profile.py 256; python tools/phase_profile.py 4096 19) 2>&1 | grep -v amdgpu.ids > $OUT/r03_phase_timeline_forward.txt
(for a in split16 w2 wino; do TG_FWD_ALGO=$a python tools/power_probe.py 65536 3; done) 2>&1 | grep -v amdgpu.ids > $OUT/r03_power_probe.txt
(TG_SP_TIMING=1 python tools/bench_selfplay.py 16 400 64 1; TG_SP_TIMING=1 python tools/bench_selfplay.py 64 400 256 1; python tools/bench_selfplay.py 1024 400 2048 1) 2>&1 | grep -v amdgpu.ids | grep "selfplay" > $OUT/r03_selfplay_rates.txt
echo legs done
