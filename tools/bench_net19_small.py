#!/usr/bin/env python3
"""19x19 forward latency of small launches: the banded kernel (a board over 4 / 2 workgroups) against the one-workgroup kernel
(TG_FWD_BANDS=0)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tamago_amd.nn.network.dual_net import DualNet
from tamago_amd import lib as tl

lib = tl.load()
net = DualNet(torch.device("cuda:0"), 19)
rs = np.random.RandomState(1)
for b in (1, 16, 64, 100, 128, 256):
    x = torch.from_numpy(rs.randint(-1, 2, size=(b, 6, 19, 19)).astype(np.float32)).cuda()
    for bands in ("", "0"):
        if bands:
            os.environ["TG_FWD_BANDS"] = bands
        else:
            os.environ.pop("TG_FWD_BANDS", None)
        for _ in range(5):
            net.forward_device(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            net.forward_device(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
        print(f"B={b:4d} {lib.tg_net_kernel_name(net.handle, b).decode():42s} {dt * 1e6:8.1f} us", flush=True)
os.environ.pop("TG_FWD_BANDS", None)
print("range fallbacks:", net.range_fallbacks())
