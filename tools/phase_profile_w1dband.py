#!/usr/bin/env python3
"""Phase timeline of dualnet_fwd_w1dband_kernel (19x19, TG_FWD_ALGO=w1dband): s_memtime stamps of pair 0's first board, both bands,
via tg_net_profile_phases - stem, the twelve layers, head convolutions; the seven stages (S, 2A, 2B, 1A, 1B, 0A, 0B) of layers 2 and 3."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TG_FWD_ALGO"] = "w1dband"
import numpy as np
import torch
from tamago_amd.nn.network.dual_net import DualNet
from tamago_amd import lib as tl
if os.environ.get("TG_EXP_LIB"):                            # an alternative build (tools/experiments/_bin/libtamago_exp<N>.so)
    tl.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiments", "_bin", f"libtamago_exp{os.environ['TG_EXP_LIB']}.so")

lib = tl.load()
net = DualNet(torch.device("cuda:0"), 19)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = torch.randint(-1, 2, (b, 6, 19, 19), device="cuda").float()
pol = torch.empty((b, 362), device="cuda")
val = torch.empty((b, 3), device="cuda")
st = np.zeros(128, dtype=np.int64)
for _ in range(2):
    tl.check(lib.tg_net_profile_phases(net.handle, x.data_ptr(), b, pol.data_ptr(), val.data_ptr(), st.ctypes.data, 128))
print("kernel:", lib.tg_net_kernel_name(net.handle, b).decode(), " batch", b)
for band in (0, 1):
    s = st[64 * band: 64 * band + 32]
    s2 = st[64 * band: 64 * band + 64]
    print(f"band {band}: board pass {s[14] - s[0]} ticks: stem {s[1] - s[0]}, head convolutions {s[14] - s[13]}")
    print("   layers:", [int(s[2 + i] - (s[1 + i] if i else s[1])) for i in range(12)])
    for layer in (2, 3):
        d = s[16 + 8 * (layer - 2): 16 + 8 * (layer - 2) + 8]
        print(f"   layer {layer} stages S 2A 2B 1A 1B 0A 0B:", [int(d[i + 1] - d[i]) for i in range(7)],
              " (72 MFMAs of 16 cycles = 1152 cycles; S: 48 / none)")
    d = s2[32:48]
    print("   conv2 layer, stage S  at slices 0 11 24 37 44 48 60 71:", [int(v - d[0]) for v in d[:8]])
    print("   conv2 layer, stage 2A at slices 0 11 24 37 44 48 60 71:", [int(v - d[8]) for v in d[8:16]])
print("start skew band 1 - band 0:", int(st[64] - st[0]), " end skew:", int(st[64 + 14] - st[14]))
