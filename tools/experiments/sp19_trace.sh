cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/lt3; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt3 -o t -- python $R/tools/bench_selfplay.py 16 100 16 0 19 > /tmp/sp19.log 2>&1
tail -1 /tmp/sp19.log
f=$(find /tmp/lt3 -name t_kernel_stats.csv | head -1); head -8 $f | cut -c1-140
