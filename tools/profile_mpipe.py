#!/usr/bin/env python3
"""Where the waves of the multi-selector PUCT kernel (select_puct_mpipe_kernel) spend their time: s_memtime
accumulators of tree 0, summed over the selector / worker waves.  TG_MPIPE_PROF=1 python tools/profile_mpipe.py"""
import os, sys
os.environ["TG_MPIPE_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tamago_amd import lib as tl
from tamago_amd.board.go_board import GoBoard
from tamago_amd.mcts.engine import SearchEngine, DeviceEvaluator
from tamago_amd.nn.network.dual_net import DualNet
size = int(sys.argv[1]) if len(sys.argv) > 1 else 9
batch = 256 if size == 9 else 64
net = DualNet(torch.device("cuda:0"), size)
eng = SearchEngine(size, 1, 1100, batch, DeviceEvaluator(net))
eng.set_root(0, GoBoard(size), 1, np.random.RandomState(0).get_state())
lib = tl.load()
eng.root_eval(False)
tl.check(lib.tg_search_profile(eng.handle, 1, None))
n = 4
for _ in range(n):
    eng.puct_batch(batch)
cyc = np.zeros(16, dtype=np.int64)
tl.check(lib.tg_search_profile(eng.handle, 0, cyc.ctypes.data))
d = n * batch
sel = ["job-ring slot wait", "root: sqrt + prior products (before the wait)", "root: wait for the predecessor",
       "root: quotients of changed children + arg-max", "below the root: wait for predecessors on the node",
       "below the root: wait for an expansion in flight", "below the root: loads + PUCB + arg-max",
       "leaf: wait for the predecessor to finish (node order)", "bookkeeping, virtual-loss stores, job hand-off"]
wrk = {9: "wait for a job", 10: "reset + replay of the path", 11: "expansion", 12: "planes + hand-back"}
print(f"ticks per descent (s_memtime; summed over waves), {cyc[13]/d:.2f} levels per descent")
tot = cyc[:9].sum()
for i, nme in enumerate(sel):
    print(f"  selector  {nme:58s} {cyc[i]/d:8.0f}  {100*cyc[i]/tot:5.1f} %")
print(f"  selector  total {tot/d:.0f} per descent over all selector waves")
wt = sum(cyc[i] for i in wrk)
for i, nme in wrk.items():
    print(f"  worker    {nme:58s} {cyc[i]/d:8.0f}  {100*cyc[i]/wt:5.1f} %")
print(f"per launch: kernel {cyc[15]/n:.0f} ticks = {cyc[15]/n/d*n:.0f} per descent; workers' root set-up (sum over workers) {cyc[14]/n:.0f}")
