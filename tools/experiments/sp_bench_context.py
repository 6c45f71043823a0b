#!/usr/bin/env python3
"""cfg-3 (16 boards) inside a process that has done what bench.py does before that leg (round 6): a 2048-tree engine stepped once, a
single-tree engine, then the self-play shard several times - per call: groups (0 = the default rule) and leaf-evals/s.
   python tools/experiments/sp_bench_context.py [groups ...]"""
import os, sys, time, tempfile, shutil
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from tamago_amd.board.go_board import GoBoard
from tamago_amd.mcts.engine import SearchEngine
from tamago_amd.nn.network.dual_net import DualNet
from tamago_amd.selfplay.worker import selfplay_shard

dev = torch.device("cuda:0")
torch.manual_seed(1234)
net = DualNet(dev, 9)
fresh = GoBoard(9, 7.0, False)
cur = torch.cuda.current_stream(dev)
if "--plain" not in sys.argv:
    ev = bench.TimedEvaluator(net)
    eng = SearchEngine(9, 2048, 1016, 256, ev, device_index=0)
    for t in range(2048):
        eng.set_root(t, fresh, 1, np.random.RandomState(t).get_state())
    plies = [np.zeros(2048, dtype=np.int64)]
    bench.run_step([(eng, cur)], plies, fresh, 1000, 256)
    torch.cuda.synchronize()
    eng.close()
    one = SearchEngine(9, 1, 1016, 256, bench.TimedEvaluator(net), device_index=0)
    one.set_root(0, fresh, 1, np.random.RandomState(7).get_state())
    p1 = [np.zeros(1, dtype=np.int64)]
    for _ in range(4):
        bench.run_step([(one, cur)], p1, fresh, 1000, 256)
    torch.cuda.synchronize()
    one.close()
for g in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [0, 0, 0, 1]:
    tmp = tempfile.mkdtemp(prefix="tg_sp_")
    t0 = time.perf_counter()
    st = selfplay_shard(tmp, net, list(range(1, 257)), 9, 400, boards=16, never_resign_flags=[True] * 256, groups=g)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    shutil.rmtree(tmp, ignore_errors=True)
    print(f"groups={g}: {st['leaf_evals'] / dt:.0f} leaf-evals/s", flush=True)
