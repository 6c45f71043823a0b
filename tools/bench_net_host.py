#!/usr/bin/env python3
"""PCIe-inclusive rate of the reference's own boundary: DualNet.inference(host tensor) ->
host tensors (tg_net_forward_host: H2D, fused forward, D2H, synchronise)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tamago_amd.nn.network.dual_net import DualNet

net = DualNet(torch.device("cuda:0"), 9)
for b in (1, 256, 4096, 65536):
    x = torch.randint(-1, 2, (b, 6, 9, 9)).float()
    for _ in range(2):
        net.inference(x)
    n = 20 if b <= 4096 else 5
    t0 = time.perf_counter()
    for _ in range(n):
        p, v = net.inference(x)
    dt = (time.perf_counter() - t0) / n
    print(f"B={b:6d}: {dt*1e6:10.1f} us per call, {b/dt:12.0f} positions/s (host tensors in and out, "
          f"{b*(6*81+85)*4/dt/1e9:.2f} GB/s over PCIe)")
