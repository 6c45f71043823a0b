"""Timing of the 9x9 forward from an alternative build of the library: tools/experiments/_bin/libtamago_<NAME>.so
(ablation builds: w1_ablation.sh / wb_ablation.sh; patched waits: patch_waits.py).  Prints the fallback counters too: a variant whose
garbage raised the range flag was timed WITH the exact kernel's redo."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))

import tamago_amd.lib as tl
name = sys.argv[1] if len(sys.argv) > 1 else "default"
if name != "default":
    tl.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", f"libtamago_{name}.so")
import torch
from tamago_amd.nn.network.dual_net import DualNet
net = DualNet(torch.device("cuda:0"), 9)
for b in (65280,):
    x = torch.randint(-1, 2, (b, 6, 9, 9), device="cuda").float()
    for _ in range(3): net.forward_device(x)
    torch.cuda.synchronize()
    n = 20 if b >= 4096 else 200
    t0 = time.perf_counter()
    for _ in range(n): net.forward_device(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"lib {name:28s} B={b:6d} {dt*1e6:9.1f} us {b/dt/1e6:7.3f} M/s   fallbacks {net.range_fallbacks()}", flush=True)
