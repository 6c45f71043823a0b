import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tamago_amd.board.go_board import GoBoard
from tamago_amd.mcts.tree import MCTSTree
from tamago_amd.mcts.time_manager import TimeManager, TimeControl
from tamago_amd.nn.network.dual_net import DualNet
net = DualNet(torch.device("cuda:0"), 19)
tree = MCTSTree(net, tree_size=8192, batch_size=64)
board = GoBoard(19, 7.0, True); color = 1
np.random.seed(0)
tm = TimeManager(TimeControl.STRICT_PLAYOUT, 1600)
for i in range(2):
    mv = tree.search_best_move(board, color, tm, {}); board.put_stone(max(mv, 0), color); color = 3 - color
t0 = time.perf_counter(); n = 6
for i in range(n):
    mv = tree.search_best_move(board, color, tm, {}); board.put_stone(max(mv, 0), color); color = 3 - color
dt = (time.perf_counter() - t0) / n
print(f"19x19 search_best_move 1600 strict visits batch 64: {dt*1e3:.2f} ms per move")
