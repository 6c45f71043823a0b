"""Where a chained single-tree search spends its time: host time inside tg_search_puct_chain vs the cursor read-back, against the per-mini-batch loop."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tamago_amd.board.go_board import GoBoard
from tamago_amd.mcts.engine import SearchEngine, DeviceEvaluator
from tamago_amd.nn.network.dual_net import DualNet
size, visits, batch = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (9, 1000, 256)
net = DualNet(torch.device("cuda:0"), size)
eng = SearchEngine(size, 1, 8192, batch, DeviceEvaluator(net))
board = GoBoard(size, 7.0, True)
batches = [batch] * (visits // batch) + ([visits % batch] if visits % batch else [])
for mode in ("loop", "chain", "loop", "chain"):
    t_root = t_search = t_collect = 0.0
    n = 12
    for it in range(n + 2):
        eng.set_root(0, board, 1, np.random.RandomState(it).get_state())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.root_eval(use_logit=False)
        eng.read_roots()
        t1 = time.perf_counter()
        if mode == "chain":
            orig = eng._collect_rng
            tc = [0.0]
            def timed():
                a = time.perf_counter(); r = orig(); tc[0] += time.perf_counter() - a; return r
            eng._collect_rng = timed
            eng.puct_chain(batches)
            eng._collect_rng = orig
        else:
            tc = [0.0]
            for k in batches:
                eng.puct_batch(k)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if it >= 2:
            t_root += t1 - t0; t_search += t2 - t1; t_collect += tc[0]
    print(f"{size}x{size} {mode:6s}: root {t_root/n*1e3:.3f} ms, search {t_search/n*1e3:.3f} ms (of which cursor read-back {t_collect/n*1e3:.3f})", flush=True)
