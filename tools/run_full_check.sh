#!/bin/bash
O=gpurun_out/${1:-r05y}; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"])
for k in ("single_tree","cfg3_selfplay_16_boards","cfg4_shard_64_boards","selfplay_1024_boards","cfg5_19x19","cfg5_19x19_trees","fp32_exact"):
    v=d.get(k,{}); print(k, v.get("value"), v.get("ms_per_move"))
PY
