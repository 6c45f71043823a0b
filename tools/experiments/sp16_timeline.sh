cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/lt2; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt2 -o t -- python $R/tools/bench_selfplay.py 16 400 64 > /tmp/sp16.log 2>&1
f=$(find /tmp/lt2 -name "*kernel_trace.csv" | head -1); python $R/tools/experiments/sp_timeline.py $f 200 > $R/gpurun_out/sp16_tl.txt
tail -1 /tmp/sp16.log
head -16 $R/gpurun_out/sp16_tl.txt
