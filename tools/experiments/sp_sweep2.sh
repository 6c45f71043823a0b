export TG_DEBUG_KNOBS=1
for w in 6 10; do for g in 2 3 4; do for cap in 0 192 224 240; do
r=$(TG_GUMBEL_WORKERS=$w TG_SP_SUBGROUPS=$g TG_SP_FWD_CAP=$cap python tools/bench_selfplay.py 16 400 64 2>&1 | grep -o '[0-9]* leaf-evals/s')
echo "workers $w sub-groups $g cap $cap: $r"; done; done; done
