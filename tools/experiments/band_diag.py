import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle.net import OracleNet, make_state_dict
from tamago_amd.nn.network.dual_net import DualNet
sd = make_state_dict(19, 3, 1.4)
net = DualNet(torch.device("cuda:0"), 19); net.load_state_dict(sd)
x = torch.from_numpy(np.random.RandomState(8).randint(-1, 2, size=(16, 6, 19, 19)).astype(np.float32))
res = {}
for name, env in (("band4", "4"), ("band2", "2"), ("plain", "0")):
    os.environ["TG_FWD_BANDS"] = env
    res[name] = [t.numpy() for t in net.inference_with_policy_logits(x)]
    again = [t.numpy() for t in net.inference_with_policy_logits(x)]
    print(name, "repeatable:", np.array_equal(res[name][0], again[0]))
ora = OracleNet(sd)
os.environ["TG_FWD_BANDS"] = "0"
pol_plain, val_plain = [t.numpy() for t in net.inference(x)]
rp, rv = [t.numpy() for t in ora.inference(x)]
print("plain vs oracle policy", np.abs(pol_plain - rp).max())
for env in ("4", "2"):
    os.environ["TG_FWD_BANDS"] = env
    p, v = [t.numpy() for t in net.inference(x)]
    print("band", env, "vs oracle policy", np.abs(p - rp).max(), "value", np.abs(v - rv).max())
for a in ("band4", "band2"):
    d = np.abs(res[a][0] - res["plain"][0])
    print(a, "vs plain: max |dlogit|", d.max(), "boards with any difference", int((d.max(axis=1) > 0).sum()), "max |logit|", np.abs(res["plain"][0]).max(),
          " entries differing", int((d > 0).sum()), "of", d.size)
print("fallbacks", net.range_fallbacks())
