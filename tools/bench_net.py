#!/usr/bin/env python3
"""Micro-benchmark of the fused DualNet forward kernel (device-resident planes)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tamago_amd.nn.network.dual_net import DualNet, random_state_dict
from tamago_amd import lib as tl

size = int(sys.argv[1]) if len(sys.argv) > 1 else 9
batches = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256, 768, 4096, 16384]
torch.manual_seed(0)
net = DualNet(torch.device("cuda:0"), size)
lib = tl.load()
flops = lib.tg_net_flops_per_position(size)
groups = os.environ.get('TG_GROUPS', '').split(',') if os.environ.get('TG_GROUPS') else [None]
for grp in groups:
  if grp: os.environ['TG_FWD_GROUP'] = grp
  for b in batches:
      x = torch.randint(-1, 2, (b, 6, size, size), device="cuda").float()
      out = (torch.empty((b, size * size + 1), device="cuda"), torch.empty((b, 3), device="cuda"))
      for _ in range(3):
          net.forward_device(x, out=out)
      torch.cuda.synchronize()
      iters = 20
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(iters):
          net.forward_device(x, out=out)
      e1.record()
      torch.cuda.synchronize()
      ms = e0.elapsed_time(e1) / iters
      print(f"S={size} B={b:6d} kernel={lib.tg_net_kernel_name(net.handle, b).decode():28s} "
            f"{ms*1e3:9.1f} us  {b/ms*1e3:12.0f} pos/s  {b*flops/ms/1e9:8.2f} TFLOP/s "
            f"({b*flops/ms/1e9/157.3*100:5.1f}% of fp32 MFMA peak)", flush=True)
