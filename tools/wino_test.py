import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle.net import OracleNet, make_state_dict
from tamago_amd.nn.network.dual_net import DualNet
sd = make_state_dict(9, 7, 1.5)
net = DualNet(torch.device("cuda:0"), 9); net.load_state_dict(sd)
ora = OracleNet(sd)
rs = np.random.RandomState(1)
x = torch.from_numpy(rs.randint(-1, 2, size=(37, 6, 9, 9)).astype(np.float32))
rp, rv = ora.inference(x)
rl, _ = ora.inference_with_policy_logits(x)
for w in ("0", "81", "82", "83"):
    os.environ["TG_FWD_WINO"] = w
    p, v = net.inference(x)
    l, _ = net.inference_with_policy_logits(x)
    print("wino", w, "policy err", float((p - rp).abs().max()), "value err", float((v - rv).abs().max()), "logit err", float((l - rl).abs().max()), flush=True)
