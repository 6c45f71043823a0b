#!/usr/bin/env python3
"""Phase timeline of the fused forward kernel (workgroup 0, s_memtime ticks = 100 MHz? no: shader clock)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tamago_amd.nn.network.dual_net import DualNet
from tamago_amd import lib as tl
lib = tl.load()
for grp in os.environ.get("TG_GROUPS", "3").split(","):
    os.environ["TG_FWD_GROUP"] = grp
    net = DualNet(torch.device("cuda:0"), 9)
    b = 65536
    x = torch.randint(-1, 2, (b, 6, 9, 9), device="cuda").float()
    pol = torch.empty((b, 82), device="cuda"); val = torch.empty((b, 3), device="cuda")
    st = np.zeros(64, dtype=np.int64)
    for _ in range(2):
        tl.check(lib.tg_net_profile_phases(net.handle, x.data_ptr(), b, pol.data_ptr(), val.data_ptr(), st.ctypes.data, 64))
    n = 1 + 1 + 12 * 2 + 1 + 1
    s = st[:n] - st[0]
    print("G =", grp, "total ticks for group 0:", s[-1])
    print(" staging+stem:", s[1])
    mf = [s[2 + 2 * i] - (s[1] if i == 0 else s[2 + 2 * i - 1]) for i in range(12)]
    ba = [s[3 + 2 * i] - s[2 + 2 * i] for i in range(12)]
    print(" layer span (epilogue of prev + MFMA loop):", mf)
    print(" barrier wait:", ba)
    print(" last epilogue:", s[26] - s[25], " heads:", s[27] - s[26])
    print(" second group stamps:", (st[n:n + 4] - st[0]).tolist())
