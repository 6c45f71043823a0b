#!/usr/bin/env python3
"""Per-move latency through the reference-shaped API (MCTSTree.search_best_move /
generate_move_with_sequential_halving), one tree, device-resident DualNet."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tamago_amd.board.go_board import GoBoard
from tamago_amd.mcts.tree import MCTSTree
from tamago_amd.mcts.time_manager import TimeManager, TimeControl
from tamago_amd.nn.network.dual_net import DualNet
net = DualNet(torch.device("cuda:0"), 9)
tree = MCTSTree(net, tree_size=4096, batch_size=256)
board = GoBoard(9, 7.0, True); color = 1
np.random.seed(0)
tm = TimeManager(TimeControl.STRICT_PLAYOUT, 1000)
for i in range(3):
    mv = tree.search_best_move(board, color, tm, {}); board.put_stone(max(mv, 0), color); color = 3 - color
t0 = time.perf_counter(); n = 20
for i in range(n):
    mv = tree.search_best_move(board, color, tm, {}); board.put_stone(max(mv, 0), color); color = 3 - color
dt = (time.perf_counter() - t0) / n
print(f"MCTSTree.search_best_move (9x9, 1000 strict visits, batch 256): {dt*1e3:.2f} ms per move = {1001/dt:.0f} leaf-evals/s")
tm2 = TimeManager(TimeControl.CONSTANT_PLAYOUT, 400)
t0 = time.perf_counter()
for i in range(n):
    mv = tree.generate_move_with_sequential_halving(board, color, tm2, True); board.put_stone(max(mv, 0), color); color = 3 - color
dt = (time.perf_counter() - t0) / n
print(f"MCTSTree.generate_move_with_sequential_halving (400 sims): {dt*1e3:.2f} ms per move = {401/dt:.0f} leaf-evals/s")
tm3 = TimeManager(TimeControl.CONSTANT_PLAYOUT, 1000)
board = GoBoard(9, 7.0, True); color = 1
t0 = time.perf_counter(); done = 0
for i in range(n):
    mv = tree.search_best_move(board, color, tm3, {}); board.put_stone(max(mv, 0), color); color = 3 - color
    done += int(tree.get_root().node_visits)
dt = (time.perf_counter() - t0) / n
print(f"MCTSTree.search_best_move (CONSTANT_PLAYOUT 1000: early stop tested after every mini-batch): {dt*1e3:.2f} ms per move, {done/n:.0f} visits per move")
