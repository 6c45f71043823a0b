"""Kernel trace target: a few 19x19 launches of 64 and 4096 boards through the band kernel (rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["TG_FWD_ALGO"] = "w1dband"
import torch
from tamago_amd.nn.network.dual_net import DualNet
net = DualNet(torch.device("cuda:0"), 19)
for b in (64, 4096):
    x = torch.randint(-1, 2, (b, 6, 19, 19), device="cuda").float()
    for _ in range(20): net.forward_device(x)
    torch.cuda.synchronize()
