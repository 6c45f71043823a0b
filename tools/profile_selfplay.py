#!/usr/bin/env python3
"""Host-side profile (cProfile) of the lock-step Gumbel self-play shard."""
import cProfile, pstats, os, sys, tempfile, shutil, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tamago_amd.nn.network.dual_net import DualNet
from tamago_amd.selfplay.worker import selfplay_shard

boards = int(sys.argv[1]) if len(sys.argv) > 1 else 64
visits = int(sys.argv[2]) if len(sys.argv) > 2 else 400
games = int(sys.argv[3]) if len(sys.argv) > 3 else boards
net = DualNet(torch.device("cuda:0"), 9)
out = tempfile.mkdtemp(prefix="sp_")
selfplay_shard(out, net, list(range(1000, 1004)), 9, 16, boards=4, never_resign_flags=[False] * 4)
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
stats = selfplay_shard(out, net, list(range(1, games + 1)), 9, visits, boards=boards,
                       never_resign_flags=[True] * games)
pr.disable()
dt = time.time() - t0
shutil.rmtree(out, ignore_errors=True)
print(f"boards={boards} visits={visits}: {stats['moves']} moves, {stats['leaf_evals']} leaf-evals in {dt:.1f} s "
      f"-> {stats['leaf_evals']/dt:.0f} leaf-evals/s (profiled)")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
