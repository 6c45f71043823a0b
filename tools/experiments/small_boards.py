import os, sys, tempfile
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tamago_amd.nn.network.dual_net import DualNet
from tamago_amd.selfplay.worker import selfplay_shard
net = DualNet(torch.device("cuda:0"), 9)
for b in (1, 2, 3, 4):
    out = tempfile.mkdtemp()
    try:
        st = selfplay_shard(out, net, list(range(1, 7)), 9, 48, boards=b, never_resign_flags=[True] * 6, groups=1, lanes=1)
        print("boards", b, "ok", st)
    except Exception as e:
        print("boards", b, "FAILED", str(e)[:600])
