"""Host-side time per step of MCTSTree.search_best_move (the reference-shaped API): where a move's milliseconds go beyond the launches."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tamago_amd.board.go_board import GoBoard
from tamago_amd.mcts.tree import MCTSTree
from tamago_amd.mcts.time_manager import TimeManager, TimeControl
from tamago_amd.nn.network.dual_net import DualNet
size, visits, batch = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (9, 1000, 256)
net = DualNet(torch.device("cuda:0"), size)
tree = MCTSTree(net, tree_size=8192 if size == 19 else 4096, batch_size=batch)
board = GoBoard(size, 7.0, True); color = 1
np.random.seed(0)
tm = TimeManager(TimeControl.STRICT_PLAYOUT, visits)
acc = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, g)
for i in range(3):
    mv = tree.search_best_move(board, color, tm, {}); board.put_stone(max(mv, 0), color); color = 3 - color
eng = tree._engine_for(board)
for n in ("set_root", "root_eval", "read_roots", "puct_chain", "puct_batch", "read_node", "num_nodes", "ensure_capacity"):
    wrap(eng, n)
wrap(tree, "_commit_rng"); wrap(tree, "search")
t0 = time.perf_counter(); n = 12
for i in range(n):
    mv = tree.search_best_move(board, color, tm, {}); board.put_stone(max(mv, 0), color); color = 3 - color
dt = (time.perf_counter() - t0) / n
print(f"{size}x{size}: {dt*1e3:.2f} ms per search_best_move; per move: " + ", ".join(f"{k} {v/n*1e3:.3f}" for k, v in acc.items()))
