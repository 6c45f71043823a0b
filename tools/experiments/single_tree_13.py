import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tamago_amd.board.go_board import GoBoard
from tamago_amd.mcts.tree import MCTSTree
from tamago_amd.mcts.time_manager import TimeManager, TimeControl
from tamago_amd.nn.network.dual_net import DualNet
torch.manual_seed(0)
net = DualNet(torch.device("cuda:0"), 13)
tree = MCTSTree(net, tree_size=8192, batch_size=128)
board = GoBoard(13, 7.0, True); color = 1
np.random.seed(0)
tm = TimeManager(TimeControl.STRICT_PLAYOUT, 1000)
mvs = []
for i in range(3):
    mv = tree.search_best_move(board, color, tm, {}); board.put_stone(max(mv, 0), color); color = 3 - color; mvs.append(mv)
t0 = time.perf_counter(); n = 8
for i in range(n):
    mv = tree.search_best_move(board, color, tm, {}); board.put_stone(max(mv, 0), color); color = 3 - color; mvs.append(mv)
dt = (time.perf_counter() - t0) / n
print(f"13x13 search_best_move 1000 visits batch 128: {dt*1e3:.2f} ms per move; moves {mvs}")
