"""search_best_move over batch sizes / visit budgets / board sizes: a cliff here means some launch fell to a one-wavefront kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tamago_amd.board.go_board import GoBoard
from tamago_amd.mcts.tree import MCTSTree
from tamago_amd.mcts.time_manager import TimeManager, TimeControl
from tamago_amd.nn.network.dual_net import DualNet
for size in (9, 19):
    torch.manual_seed(0)
    net = DualNet(torch.device("cuda:0"), size)
    for batch, visits in ((16, 1000), (64, 1000), (256, 1000), (512, 2000), (1024, 4000), (2048, 8000)):
        tree = MCTSTree(net, tree_size=65536, batch_size=batch)
        board = GoBoard(size, 7.0, True); color = 1
        np.random.seed(0)
        tm = TimeManager(TimeControl.STRICT_PLAYOUT, visits)
        for i in range(2):
            mv = tree.search_best_move(board, color, tm, {}); board.put_stone(max(mv, 0), color); color = 3 - color
        t0 = time.perf_counter(); n = 4
        for i in range(n):
            mv = tree.search_best_move(board, color, tm, {}); board.put_stone(max(mv, 0), color); color = 3 - color
        dt = (time.perf_counter() - t0) / n
        print(f"{size}x{size} batch {batch:5d} visits {visits:5d}: {dt*1e3:8.2f} ms per move = {visits/dt/1e3:7.1f} k leaf-evals/s", flush=True)
