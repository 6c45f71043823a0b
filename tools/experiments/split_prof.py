#!/usr/bin/env python3
"""Who waits for whom inside select_puct_split_kernel (one tree): s_memtime stamps of tree 0 from the -DTG_SPLIT_PROF build
(tools/experiments/split_prof.sh).   python tools/experiments/split_prof.py [9|19] [launches]"""
import os, sys
os.environ["TG_MPIPE_PROF"] = "1"
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np, torch
from tamago_amd import lib as tl
tl.LIB_PATH = os.path.join(ROOT, "tools/experiments/_bin/libtamago_splitprof.so")
from tamago_amd.board.go_board import GoBoard
from tamago_amd.mcts.engine import SearchEngine, DeviceEvaluator
from tamago_amd.nn.network.dual_net import DualNet
size = int(sys.argv[1]) if len(sys.argv) > 1 else 9
n = int(sys.argv[2]) if len(sys.argv) > 2 else (4 if size == 9 else 25)
batch = 256 if size == 9 else 64
net = DualNet(torch.device("cuda:0"), size)
eng = SearchEngine(size, 1, n * batch + 100, batch, DeviceEvaluator(net))
eng.set_root(0, GoBoard(size), 1, np.random.RandomState(0).get_state())
lib = tl.load()
eng.root_eval(False)
tl.check(lib.tg_search_profile(eng.handle, 1, None))
names = ["clerk: loop", "clerk: wait for a free slot", "allocator: loop", "allocator: wait for the next leaf",
         "node owner 1: busy", "node owner 1: steps", "shipper 0: loop", "worker (1,0): wait for a job", "worker (1,0): header + reset + replay",
         "worker (1,0): candidates / prior / done tag", "worker (1,0): planes", "-", "-", "-", "-", "selecting half: start to end"]
print(f"{size}x{size}, {batch} descents per launch; ticks of s_memtime (100 MHz: 1 tick = 10 ns)")
for it in range(n):
    eng.puct_batch(batch)
    cyc = np.zeros(16, dtype=np.int64)
    tl.check(lib.tg_search_profile(eng.handle, 1, cyc.ctypes.data))
    last_worker = int(cyc[13] - cyc[12])
    if it in (0, 1, n // 2, n - 1):
        print(f"launch {it}: last worker done {last_worker / 100:.1f} us after the selecting half started; selecting half {cyc[15] / 100:.1f} us")
        print(f"    {'chooser: loop':48s} {cyc[14] / 100:9.1f} us")
        if cyc[11]:
            print(f"    node owners: {cyc[11]} steps, {cyc[14]} on the node the owner handled last ({100 * cyc[14] / cyc[11]:.0f} %)")
        for i in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10):
            print(f"    {names[i]:48s} {cyc[i] / (1 if i == 5 else 100):9.1f}" + ("" if i == 5 else " us"))
