import os, sys, tempfile, time
sys.path.insert(0, "/root/repo")
import torch
from tamago_amd.nn.network.dual_net import DualNet
from tamago_amd.selfplay.worker import selfplay_shard
from oracle.net import make_state_dict
net = DualNet(torch.device("cuda:0"), 9); net.load_state_dict(make_state_dict(9, 23, 1.5))
idx = list(range(301, 301 + 40)); flags = [i % 5 != 0 for i in idx]
ref = None
for chain, sub in ((0, 1), (1, 1), (1, 2), (1, 3), (1, 4)):
    os.environ["TG_SP_CHAIN"] = str(chain); os.environ["TG_SP_SUBGROUPS"] = str(sub)
    d = tempfile.mkdtemp()
    t0 = time.time()
    st = selfplay_shard(d, net, idx, 9, 400, boards=16, never_resign_flags=flags)
    dt = time.time() - t0
    texts = [open(f"{d}/{i}.sgf").read() for i in idx]
    if ref is None: ref = (st, texts)
    print(chain, sub, st, f"{st['leaf_evals']/dt:.0f}/s", "same" if (st, texts) == ref else "DIFFERENT", flush=True)
