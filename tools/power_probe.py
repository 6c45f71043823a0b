#!/usr/bin/env python3
"""Board power and shader clock while one forward kernel runs back to back (rocm-smi samples from a side thread):
is the kernel's rate bounded by the power budget (DVFS) rather than by its instruction stream?
   TG_FWD_ALGO=s32|split16|wino python tools/power_probe.py [batch] [seconds]"""
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tamago_amd.nn.network.dual_net import DualNet
from tamago_amd import lib as tl

b = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
zero = os.environ.get("TG_PROBE_ZERO") is not None
torch.manual_seed(0)
net = DualNet(torch.device("cuda:0"), 9)
lib = tl.load()
x = torch.zeros((b, 6, 9, 9), device="cuda") if zero else torch.randint(-1, 2, (b, 6, 9, 9), device="cuda").float()
out = (torch.empty((b, 82), device="cuda"), torch.empty((b, 3), device="cuda"))
samples = []
stop = False


def sample():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = d[sorted(d)[0]]
            samples.append({k: v for k, v in card.items() if "ower" in k or "sclk" in k.lower()})
        except Exception as exc:
            samples.append({"error": repr(exc)})
        time.sleep(0.15)


for _ in range(3):
    net.forward_device(x, out=out)
torch.cuda.synchronize()
th = threading.Thread(target=sample)
th.start()
t0 = time.perf_counter()
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.perf_counter() - t0 < secs:
    for _ in range(10):
        net.forward_device(x, out=out)
    n += 10
    torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
stop = True
th.join()
ms = e0.elapsed_time(e1) / n
print(f"{os.environ.get('TG_FWD_ALGO', 'default'):8s} {lib.tg_net_kernel_name(net.handle, b).decode():40s} zero_input={zero} "
      f"{ms:8.3f} ms per launch  {b / ms / 1e3:6.3f} M pos/s")
for s in samples[2:10]:
    print("   ", s)
