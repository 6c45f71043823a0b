"""Helper of test_gpu_selfplay.py::test_gumbel_kernel_variants_write_the_same_games (run as a script: the Gumbel
selection kernel variant is chosen from the environment once per process).  Plays a few lock-step self-play games
with a seeded random-init DualNet and prints a digest of the SGF files: argv = boards, games, visits[, board size]."""
import hashlib, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tamago_amd.nn.network.dual_net import DualNet
from tamago_amd.selfplay.worker import selfplay_shard
boards, games, visits = (int(v) for v in sys.argv[1:4])
size = int(sys.argv[4]) if len(sys.argv) > 4 else 9
torch.manual_seed(33)
net = DualNet(torch.device("cuda:0"), size)
idx = list(range(1, games + 1))
flags = [i % 4 == 0 for i in idx]
h = hashlib.sha256()
with tempfile.TemporaryDirectory() as d:
    stats = selfplay_shard(d, net, idx, size, visits, boards=boards, never_resign_flags=flags)
    for i in idx:
        h.update(open(os.path.join(d, f"{i}.sgf"), "rb").read())
print(h.hexdigest()[:16], stats["games"], stats["moves"])
