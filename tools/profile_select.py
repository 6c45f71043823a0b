#!/usr/bin/env python3
"""Per-phase cycle breakdown of the PUCT selection kernel (tree 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tamago_amd import lib as tl
from tamago_amd.board.go_board import GoBoard
from tamago_amd.mcts.engine import SearchEngine, DeviceEvaluator
from tamago_amd.nn.network.dual_net import DualNet
trees = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = DualNet(torch.device("cuda:0"), 9)
eng = SearchEngine(9, trees, 1100, 256, DeviceEvaluator(net))
for t in range(trees):
    eng.set_root(t, GoBoard(9), 1, np.random.RandomState(t).get_state())
lib = tl.load()
eng.root_eval(False)
tl.check(lib.tg_search_profile(eng.handle, 1, None))
for _ in range(4):
    eng.puct_batch(256)
cyc = np.zeros(16, dtype=np.int64)
tl.check(lib.tg_search_profile(eng.handle, 0, cyc.ctypes.data))
names = ["board reset", "PUCB select", "put_stone", "edge bookkeeping", "expansion", "planes+queue", "-", "levels"]
tot = cyc[:6].sum()
print(f"trees={trees}: {tot/1024:.0f} cycles per descent ({tot/1024/2.4e3:.2f} us at 2.4 GHz), {cyc[7]/1024:.2f} levels per descent")
for n, c in zip(names[:6], cyc[:6]):
    print(f"  {n:18s} {c/1024:9.0f} cycles/descent  {100*c/tot:5.1f} %")
print(f"  expansion detail: candidates {cyc[8]/1024:.0f}, dirichlet sum {cyc[9]/1024:.0f}, node init + sync {cyc[10]/1024:.0f} cycles/descent")
