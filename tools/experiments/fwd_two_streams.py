#!/usr/bin/env python3
"""Forward launches of one network on two streams at once (the shape a self-play move's sub-groups give them): same bits as alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tamago_amd.nn.network.dual_net import DualNet
dev = torch.device("cuda:0")
torch.manual_seed(3)
net = DualNet(dev, 9)
sizes = [530, 96, 1600, 257, 768, 5, 300]
xs = [torch.randint(-1, 2, (b, 6, 9, 9), device=dev).float() for b in sizes]
want = [tuple(t.clone() for t in net.forward_device(x, True)) for x in xs]
torch.cuda.synchronize()
s2 = torch.cuda.Stream()
bad = 0
for it in range(300):
    outs = []
    for i, x in enumerate(xs):
        if (i + it) % 2:
            with torch.cuda.stream(s2):
                outs.append(net.forward_device(x, True))
        else:
            outs.append(net.forward_device(x, True))
    torch.cuda.synchronize()
    for (p, v), (wp, wv) in zip(outs, want):
        if not (torch.equal(p, wp) and torch.equal(v, wv)):
            bad += 1
print("iterations 300, mismatching launches:", bad)
