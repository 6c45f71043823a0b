#!/usr/bin/env python3
"""Throughput of the lock-step Gumbel self-play shard (BASELINE.json configs 3 / 4): argv = boards, visits, games, groups (0: default), board size (9)."""
import os, sys, time, tempfile, shutil
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")          # before torch loads the HIP runtime (tamago_amd/__init__.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tamago_amd.nn.network.dual_net import DualNet
from tamago_amd.selfplay.worker import selfplay_shard

boards = int(sys.argv[1]) if len(sys.argv) > 1 else 16
visits = int(sys.argv[2]) if len(sys.argv) > 2 else 400
games = int(sys.argv[3]) if len(sys.argv) > 3 else boards
groups = int(sys.argv[4]) if len(sys.argv) > 4 else 0
size = int(sys.argv[5]) if len(sys.argv) > 5 else 9
torch.manual_seed(0)
net = DualNet(torch.device("cuda:0"), size)
out = tempfile.mkdtemp(prefix="sp_")
# warm-up: one short batch of games
selfplay_shard(out, net, list(range(1000, 1000 + min(boards, 4))), size, 16, boards=min(boards, 4),
               never_resign_flags=[False] * min(boards, 4))
t0 = time.time()
stats = selfplay_shard(out, net, list(range(1, games + 1)), size, visits, boards=boards,
                       never_resign_flags=[True] * games, groups=groups)
dt = time.time() - t0
shutil.rmtree(out, ignore_errors=True)
print(f"selfplay {size}x{size} boards={boards} groups={groups or 'auto'} visits={visits}: {stats['games']} games, {stats['moves']} moves, "
      f"{stats['leaf_evals']} leaf-evals in {dt:.1f} s -> {stats['leaf_evals']/dt:.0f} leaf-evals/s, "
      f"{stats['games']/dt*3600:.0f} games/hour")
