import cProfile, pstats, os, sys, tempfile, shutil, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tamago_amd.nn.network.dual_net import DualNet
from tamago_amd.selfplay.worker import selfplay_shard
boards = int(sys.argv[1]); groups = int(sys.argv[2])
net = DualNet(torch.device("cuda:0"), 9)
out = tempfile.mkdtemp(prefix="sp_")
selfplay_shard(out, net, list(range(1000, 1004)), 9, 16, boards=4, never_resign_flags=[False] * 4)
pr = cProfile.Profile()
t0 = time.time(); pr.enable()
stats = selfplay_shard(out, net, list(range(1, 2 * boards + 1)), 9, 400, boards=boards, never_resign_flags=[True] * (2 * boards), groups=groups)
pr.disable(); dt = time.time() - t0
shutil.rmtree(out, ignore_errors=True)
print(f"boards={boards} groups={groups}: {stats['leaf_evals']/dt:.0f} leaf-evals/s in {dt:.1f}s")
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
