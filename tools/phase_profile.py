#!/usr/bin/env python3
"""Phase timeline of the fused forward kernel: s_memtime stamps (shader-clock ticks) of workgroup 0
on its first board group, via tg_net_profile_phases.

Winograd kernel (9x9 default): wave 0 (half 0, three row-tiles) and wave 4 (half 1, two row-tiles)
share SIMD 0; for each of the 12 tower layers the time this wave worked and the time it then
waited at the layer barrier.  Direct kernel (TG_FWD_ALGO=direct): wave 0 only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tamago_amd.nn.network.dual_net import DualNet
from tamago_amd import lib as tl

lib = tl.load()
size = int(sys.argv[2]) if len(sys.argv) > 2 else 9
net = DualNet(torch.device("cuda:0"), size)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
if size == 9 and b > 768:
    b -= b % 768        # whole rounds of 3-board workgroups: a ragged tail would be a second launch (1-board workgroups)
                        # whose stamps overwrite the first one's
x = torch.randint(-1, 2, (b, 6, size, size), device="cuda").float()
pol = torch.empty((b, size * size + 1), device="cuda")
val = torch.empty((b, 3), device="cuda")
st = np.zeros(128, dtype=np.int64)
for _ in range(2):
    tl.check(lib.tg_net_profile_phases(net.handle, x.data_ptr(), b, pol.data_ptr(), val.data_ptr(),
                                       st.ctypes.data, 128))
name = lib.tg_net_kernel_name(net.handle, b).decode()
print("kernel:", name)
if "w2" in name:
    # two stamping waves (wave 0 and the last wave of workgroup 0); per layer: MFMA loop done, barrier passed (= every
    # wave's loop done), epilogue stored + barrier passed
    for label, base in (("wave 0", 0), ("last wave", 64)):
        s = st[base:base + 42] - st[base]
        print(f"{label}: group total {s[41]} ticks; staging + im2col + split of the input {s[1]}")
        print("  MFMA loop (stem, then 12 layers)   :", [int(s[2 + 3 * i] - s[1 + 3 * i]) for i in range(13)])
        print("  wait for the other waves (barrier) :", [int(s[3 + 3 * i] - s[2 + 3 * i]) for i in range(13)])
        print("  epilogue + barrier                 :", [int(s[4 + 3 * i] - s[3 + 3 * i]) for i in range(13)])
        print("  heads                              :", int(s[41] - s[40]))
    print("  heads detail (wave 0): 1x1 convs", int(st[44] - st[40]), "| FC weights landed + barrier", int(st[45] - st[44]),
          "| FCs", int(st[46] - st[45]), "| softmax + stores", int(st[41] - st[46]))
    print("  start skew of the last wave vs wave 0:", int(st[64] - st[0]))
elif "split" in name:
    s = st[:29] - st[0]
    print(f"group total {s[28]} ticks; staging + im2col + split of the input {s[1]}")
    print("  MFMA loop (stem, then 12 layers):", [int(s[2 + 2 * i] - s[1 + 2 * i]) for i in range(13)])
    print("  barrier + epilogue + barrier    :", [int(s[3 + 2 * i] - s[2 + 2 * i]) for i in range(13)])
    if size == 9:
        print("  heads                           :", int(s[28] - s[27]), " (1x1 convs", int(st[40] - st[27]),
              "| FC weights landed + barrier", int(st[41] - st[40]), "| FCs", int(st[42] - st[41]), "| softmax + stores",
              int(st[28] - st[42]), ")")
    else:
        print("  heads                           :", int(s[28] - s[27]))
elif "wino" in name:
    for label, s in (("half 0 (wave 0)", st[:64]), ("half 1 (wave 4)", st[64:])):
        s = s - st[0]
        print(f"{label}: group total {s[26]} ticks; staging + stem {s[1]}")
        print("  layer work  :", [int(s[2 + 2 * i] - s[1 + 2 * i]) for i in range(12)])
        print("  barrier wait:", [int(s[3 + 2 * i] - s[2 + 2 * i]) for i in range(12)])
        print("  heads       :", int(s[26] - s[25]))
else:
    n = 1 + 1 + 12 * 2 + 1 + 1
    s = st[:n] - st[0]
    print("total ticks for group 0:", s[-1], " staging + stem:", s[1])
    print(" layer span (epilogue of prev + MFMA loop):",
          [int(s[2 + 2 * i] - (s[1] if i == 0 else s[2 + 2 * i - 1])) for i in range(12)])
    print(" barrier wait:", [int(s[3 + 2 * i] - s[2 + 2 * i]) for i in range(12)])
    print(" last epilogue:", int(s[26] - s[25]), " heads:", int(s[27] - s[26]))
