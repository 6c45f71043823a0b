cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 16 64; do
python $R/tools/bench_selfplay.py $B 400 $((2*B)) 2>&1 | grep selfplay
rm -rf /tmp/sp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o t -- python $R/tools/bench_selfplay.py $B 400 $((2*B)) 2>&1 | grep selfplay
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/sp/**/t_kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("   ", r['Name'][27:90], r['Calls'], "%.1f us"%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
done
