#!/usr/bin/env python3
"""Bit-identity of the one-board and the three-board variant of the default 9x9 forward kernel on Go-like planes (0 / 1 stone
planes, one-hot previous move, constant colour plane) with the hot test network of the self-play scheme test."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from oracle.net import make_state_dict
from tamago_amd.nn.network.dual_net import DualNet

net = DualNet(torch.device("cuda:0"), 9)
seed, gain = (int(sys.argv[1]), float(sys.argv[2])) if len(sys.argv) > 2 else (23, 1.5)
net.load_state_dict(make_state_dict(9, seed, gain))
rs = np.random.RandomState(5)
n = 1200
x = np.zeros((n, 6, 81), dtype=np.float32)
for b in range(n):
    dens = rs.uniform(0.0, 0.9)
    cells = rs.choice(3, size=81, p=[1 - dens, dens / 2, dens / 2])
    x[b, 0] = cells == 0
    x[b, 1] = cells == 1
    x[b, 2] = cells == 2
    if rs.rand() < 0.1:
        x[b, 4] = 1.0
    elif dens > 0:
        x[b, 3, rs.randint(81)] = 1.0
    x[b, 5] = 1.0 if rs.rand() < 0.5 else -1.0
x = torch.from_numpy(x.reshape(n, 6, 9, 9)).cuda()
big_p, big_v = [t.cpu().numpy() for t in net.forward_device(x[:768])]        # one round of three-board groups
bad = []
for lo in range(0, 768, 96):
    p, v = [t.cpu().numpy() for t in net.forward_device(x[lo:lo + 96])]      # one-board workgroups
    d = np.nonzero((p.view(np.uint32) != big_p[lo:lo + 96].view(np.uint32)).any(axis=1) | (v.view(np.uint32) != big_v[lo:lo + 96].view(np.uint32)).any(axis=1))[0]
    bad += [lo + int(i) for i in d]
print("range fallbacks:", net.range_fallbacks())
print(f"{len(bad)} of 768 positions differ between the variants", bad[:20])
for i in bad[:5]:
    p1, _ = [t.cpu().numpy() for t in net.forward_device(x[i:i + 1])]
    print(i, "max |dp|", np.abs(p1[0] - big_p[i]).max(), "stones", int(x[i, 1].sum().item() + x[i, 2].sum().item()))
# the ragged-tail path: 768 positions as three-board groups + 96 behind them as one-board workgroups, back to back
ref_p, ref_v = [t.cpu().numpy() for t in net.forward_device(x[:768])]
tail_p = np.concatenate([net.forward_device(x[768 + lo:768 + lo + 1])[0].cpu().numpy() for lo in range(96)])
worst = 0
for rep in range(40):
    outs = [net.forward_device(x[:864]) for _ in range(3)]
    torch.cuda.synchronize()
    for p, v in outs:
        p = p.cpu().numpy()
        nbad_head = int((p[:768].view(np.uint32) != ref_p.view(np.uint32)).any(axis=1).sum())
        nbad_tail = int((p[768:].view(np.uint32) != tail_p.view(np.uint32)).any(axis=1).sum())
        if nbad_head or nbad_tail:
            worst += 1
            if worst < 6:
                print(f"rep {rep}: {nbad_head} head / {nbad_tail} tail positions differ; max |dp| tail {np.abs(p[768:] - tail_p).max():.3e}")
print("ragged launches with differences:", worst, "of 120")
# two streams at once: each forwards its own ragged batch over and over; every result against the single-stream one
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
xa, xb = x[:864], x[300:300 + 808].contiguous()
ra = [t.cpu().numpy() for t in net.forward_device(xa)]
rb = [t.cpu().numpy() for t in net.forward_device(xb)]
torch.cuda.synchronize()
outs_a, outs_b = [], []
for rep in range(60):
    with torch.cuda.stream(s1):
        outs_a.append(net.forward_device(xa))
    with torch.cuda.stream(s2):
        outs_b.append(net.forward_device(xb))
        outs_b.append(net.forward_device(xb[:40]))
torch.cuda.synchronize()
na = sum(int((p.cpu().numpy().view(np.uint32) != ra[0].view(np.uint32)).any()) for p, v in outs_a)
nb = sum(int((p.cpu().numpy().view(np.uint32) != (rb[0] if p.shape[0] == 808 else rb[0][:40]).view(np.uint32)).any()) for p, v in outs_b)
print(f"two streams: {na} of {len(outs_a)} + {nb} of {len(outs_b)} launches differ from the single-stream results")
for p, v in outs_a:
    d = np.nonzero((p.cpu().numpy().view(np.uint32) != ra[0].view(np.uint32)).any(axis=1))[0]
    if len(d):
        print("  stream 1 positions", d[:12], "of 864 (tail = 768 ..)")
        break
for p, v in outs_b:
    ref = rb[0] if p.shape[0] == 808 else rb[0][:40]
    d = np.nonzero((p.cpu().numpy().view(np.uint32) != ref.view(np.uint32)).any(axis=1))[0]
    if len(d):
        print("  stream 2 positions", d[:12], "of", p.shape[0], "(tail = 768 ..)")
        break
