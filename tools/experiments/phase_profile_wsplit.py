#!/usr/bin/env python3
"""Phase timeline of dualnet_fwd_wsplit_kernel (TG_FWD_ALGO=wsplit): s_memtime stamps of workgroup 0 / wave 0 on its first
board group via tg_net_profile_phases - per layer, and per row tile inside layers 2 (conv1) and 3 (conv2)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TG_FWD_ALGO"] = "wsplit"
import numpy as np
import torch
from tamago_amd.nn.network.dual_net import DualNet
from tamago_amd import lib as tl

lib = tl.load()
net = DualNet(torch.device("cuda:0"), 9)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 65280
if b > 768:
    b -= b % 768
x = torch.randint(-1, 2, (b, 6, 9, 9), device="cuda").float()
pol = torch.empty((b, 82), device="cuda")
val = torch.empty((b, 3), device="cuda")
st = np.zeros(128, dtype=np.int64)
for _ in range(2):
    tl.check(lib.tg_net_profile_phases(net.handle, x.data_ptr(), b, pol.data_ptr(), val.data_ptr(), st.ctypes.data, 128))
print("kernel:", lib.tg_net_kernel_name(net.handle, b).decode(), " batch", b)
s = st[:16] - st[0]
print(f"group total {s[15]} ticks: staging {s[1]}, stem {s[2] - s[1]}, heads {s[15] - s[14]}")
print("  layers:", [int(s[3 + i] - s[2 + i]) for i in range(12)])
nrt = 5 if b > 256 else 2
names = ["phase A (MFMA kc0 | transform kc1 | previous tail + epilogue)", "phase B (MFMA kc1 | transform next kc0 | Z)", "tail + barrier (last row tile)", "own epilogue (last row tile)"]
for layer in (2, 3):
    base = 40 + 35 * (layer - 2)
    print(f"layer {layer} ({'conv1' if layer % 2 == 0 else 'conv2 + residual'}), per row tile:")
    for rt in range(nrt):
        d = st[base + 7 * rt: base + 7 * rt + 7]
        print(f"  rt {rt}: total {int(d[4] - d[0]):5d} |", " | ".join(f"{n} {int(d[i + 1] - d[i])}" for i, n in enumerate(names)))
    if nrt > 1:
        gaps = [int(st[base + 7 * (rt + 1)] - st[base + 7 * rt + 4]) for rt in range(nrt - 1)]
        print("  between row tiles:", gaps)
