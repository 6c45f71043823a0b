#!/usr/bin/env python3
"""Banded 19x19 launches from two streams at once (256 workgroups each: they cannot both be resident): every result equals the
single-stream one, nothing is redone by the exact kernel, nothing waits for a timeout."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tamago_amd.nn.network.dual_net import DualNet
net = DualNet(torch.device("cuda:0"), 19)
x = torch.from_numpy(np.random.RandomState(2).randint(-1, 2, size=(128, 6, 19, 19)).astype(np.float32)).cuda()
xa, xb = x[:64], x[64:].contiguous()
ra, rb = net.forward_device(xa), net.forward_device(xb)
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
t0 = time.perf_counter()
oa, ob = [], []
for _ in range(100):
    with torch.cuda.stream(s1):
        oa.append(net.forward_device(xa))
    with torch.cuda.stream(s2):
        ob.append(net.forward_device(xb))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
bad = sum(int(not (torch.equal(p, ra[0]) and torch.equal(v, ra[1]))) for p, v in oa) + sum(int(not (torch.equal(p, rb[0]) and torch.equal(v, rb[1]))) for p, v in ob)
print(f"200 banded launches on two streams: {dt * 1e3:.1f} ms ({dt / 200 * 1e6:.1f} us each), {bad} differ, fallbacks {net.range_fallbacks()}")
