#!/usr/bin/env python3
"""Forward-only rate of the default 9x9 kernel over launch sizes (whole rounds of 768 positions, ragged remainders)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from tamago_amd.nn.network.dual_net import DualNet
from tamago_amd import lib as tl

lib = tl.load()
net = DualNet(torch.device("cuda:0"), 9)
rs = np.random.RandomState(1)
sizes = [int(a) for a in sys.argv[1:]] or [768 * 85, 65536, 768 * 86, 768 * 85 + 300, 768 * 128, 131072, 196608, 6400, 6144, 768 * 8 + 256, 100000]
for b in sizes:
    x = torch.from_numpy(rs.randint(-1, 2, size=(b, 6, 9, 9)).astype(np.float32)).cuda()
    for _ in range(3):
        net.forward_device(x)
    torch.cuda.synchronize()
    n = 20 if b >= 65536 else 100
    t0 = time.perf_counter()
    for _ in range(n):
        net.forward_device(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    rounds = b / 768.0
    print(f"B={b:7d} ({rounds:7.2f} rounds) {lib.tg_net_kernel_name(net.handle, b).decode():40s} {dt * 1e6:9.1f} us  {dt * 1e6 / rounds:6.1f} us/round  {b / dt / 1e6:6.3f} M positions/s", flush=True)
