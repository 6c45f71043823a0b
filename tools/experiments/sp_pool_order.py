#!/usr/bin/env python3
"""Does the ORDER in which HIP streams are created decide how two self-play lanes perform?  (round 6)
   python tools/experiments/sp_pool_order.py <pool-first|pool-late|own> [boards]"""
import os, sys, time, tempfile, shutil
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1]
boards = int(sys.argv[2]) if len(sys.argv) > 2 else 16
if mode == "own":
    os.environ["TG_SP_LANE_STREAMS"] = "own"
import torch
from tamago_amd.nn.network.dual_net import DualNet
from tamago_amd.selfplay.worker import selfplay_shard
net = DualNet(torch.device("cuda:0"), 9)
if mode == "pool-first":
    keep = torch.cuda.Stream()                      # torch creates its pool of 32 streams here, before any library stream
out = tempfile.mkdtemp(prefix="sp_")
if mode == "warm2":
    os.environ["TG_SP_LANES"] = "2"
    selfplay_shard(out, net, list(range(1000, 1004)), 9, 16, boards=4, never_resign_flags=[False] * 4)
elif mode == "warm1-stream":
    os.environ["TG_SP_LANES"] = "1"
    with torch.cuda.stream(torch.cuda.Stream()):
        selfplay_shard(out, net, list(range(1000, 1004)), 9, 16, boards=4, never_resign_flags=[False] * 4)
elif mode == "nowarm":
    pass
else:
    os.environ["TG_SP_LANES"] = "1"
    selfplay_shard(out, net, list(range(1000, 1004)), 9, 16, boards=4, never_resign_flags=[False] * 4)     # one group: library streams
os.environ["TG_SP_LANES"] = "2"
t0 = time.time()
stats = selfplay_shard(out, net, list(range(1, 257)), 9, 400, boards=boards, never_resign_flags=[True] * 256)
dt = time.time() - t0
shutil.rmtree(out, ignore_errors=True)
print(f"{mode}: {stats['leaf_evals'] / dt:.0f} leaf-evals/s")
