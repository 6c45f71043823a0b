"""Timing of the 19x19 band kernel from an alternative build of the library (experiments: tools/experiments/_bin/libtamago_exp<N>.so)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["TG_FWD_ALGO"] = "w1dband"
import tamago_amd.lib as tl
if len(sys.argv) > 1:
    tl.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", f"libtamago_exp{sys.argv[1]}.so")
import torch
from tamago_amd.nn.network.dual_net import DualNet
net = DualNet(torch.device("cuda:0"), 19)
for b in (64, 4096):
    x = torch.randint(-1, 2, (b, 6, 19, 19), device="cuda").float()
    for _ in range(3): net.forward_device(x)
    torch.cuda.synchronize()
    n = 20 if b >= 4096 else 200
    t0 = time.perf_counter()
    for _ in range(n): net.forward_device(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"lib {sys.argv[1] if len(sys.argv) > 1 else 'default'} B={b} {dt*1e6:.1f} us {b/dt/1e6:.3f} M/s", flush=True)
