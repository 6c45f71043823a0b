#!/usr/bin/env python3
"""Where the waves of the node-owner PUCT kernel (select_puct_owner_kernel) spend their time: s_memtime accumulators
of tree 0.  TG_SELECT_OWNER=1 TG_MPIPE_PROF=1 python tools/profile_owner.py [size]"""
import os, sys
os.environ["TG_MPIPE_PROF"] = "1"
os.environ.setdefault("TG_SELECT_OWNER", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tamago_amd import lib as tl
from tamago_amd.board.go_board import GoBoard
from tamago_amd.mcts.engine import SearchEngine, DeviceEvaluator
from tamago_amd.nn.network.dual_net import DualNet
size = int(sys.argv[1]) if len(sys.argv) > 1 else 9
batch = 256 if size == 9 else 64
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # mini-batches before the measured ones (deeper trees)
net = DualNet(torch.device("cuda:0"), size)
eng = SearchEngine(size, 1, (warm + 5) * batch + 100, batch, DeviceEvaluator(net))
eng.set_root(0, GoBoard(size), 1, np.random.RandomState(0).get_state())
lib = tl.load()
eng.root_eval(False)
for _ in range(warm):
    eng.puct_batch(batch)
tl.check(lib.tg_search_profile(eng.handle, 1, None))
n = 4
for _ in range(n):
    eng.puct_batch(batch)
cyc = np.zeros(16, dtype=np.int64)
tl.check(lib.tg_search_profile(eng.handle, 0, cyc.ctypes.data))
d = n * batch
print(f"ticks per descent (s_memtime), {cyc[4]/d:.2f} levels below the root per descent")
print(f"  root owner    : busy {cyc[0]/d:7.0f}   waiting for a free slot {cyc[1]/d:7.0f}")
print(f"  node owners   : busy {cyc[2]/d:7.0f}   polling {cyc[3]/d:7.0f}   (sum over the waves; {cyc[2]/max(cyc[4],1):.0f} per visit)")
print(f"  allocator     : busy {cyc[5]/d:7.0f}   waiting for the next leaf {cyc[6]/d:7.0f}")
print(f"  workers       : busy {cyc[7]/d:7.0f}   waiting for a job {cyc[8]/d:7.0f}   (sum over the waves)")
print(f"per launch: kernel {cyc[15]/n:.0f} ticks = {cyc[15]/d:.0f} per descent")
