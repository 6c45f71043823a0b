#!/usr/bin/env python3
"""Where the selector wave of the Gumbel selection kernel (select_gumbel_pipe_kernel) spends a move's four phases:
s_memtime accumulators of tree 0.  python tools/profile_gumbel.py [moves]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tamago_amd import lib as tl
from tamago_amd.board.go_board import GoBoard
from tamago_amd.mcts.tree import MCTSTree
from tamago_amd.mcts.time_manager import TimeManager, TimeControl
from tamago_amd.nn.network.dual_net import DualNet
moves = int(sys.argv[1]) if len(sys.argv) > 1 else 24
net = DualNet(torch.device("cuda:0"), 9)
tree = MCTSTree(net, tree_size=4096)
board = GoBoard(9, 7.0, True)
lib = tl.load()
np.random.seed(0)
color = 1
cyc = np.zeros(16, dtype=np.int64)
n = 0
for m in range(moves):
    mv = tree.generate_move_with_sequential_halving(board, color, TimeManager(TimeControl.CONSTANT_PLAYOUT, 400), True)
    if m == 3:
        tl.check(lib.tg_search_profile(tree._engine.handle, 1, None))          # counters on from the fifth move
    elif m > 3:
        n += 1
    board.put_stone(mv, color); color = 3 - color
tl.check(lib.tg_search_profile(tree._engine.handle, 0, cyc.ctypes.data))
ph = 4 * n
names = {0: "set-up (root into registers, ranking)", 1: "root choices of the phase (entries, schedule)",
         7: "its share of the entries' walks", 3: "waiting for the other waves' shares", 8: "nodes / EXPAND jobs handed out", 2: "first descents (step into the new node, LEAF job)",
         4: "(waiting for a free job slot, inside the above)", 5: "write-back"}
print(f"selector wave, ticks per phase over {ph} phases ({cyc[6]/ph:.1f} first descents per phase; the repeats are the workers')")
for i, nm in names.items():
    print(f"  {nm:50s} {cyc[i]/ph:9.0f}")
L = max(cyc[12], 1)
print(f"  {cyc[12]} launches, {cyc[14]/L:.1f} threshold levels walked one by one per launch")
print(f"first worker wave, ticks per launch: waiting for the selector's go {cyc[9]/L:.0f}, its entries {cyc[10]/L:.0f}, scheduled copies {cyc[11]/L:.0f}; kernel start to its end {cyc[13]/L:.0f}")
print(f"  per first descent {cyc[2]/max(cyc[6],1):.0f}; selector done after {cyc[15]/ph:.0f} ticks per phase")
