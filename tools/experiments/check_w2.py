#!/usr/bin/env python3
"""Bring-up / regression check of the two-waves-per-SIMD forward kernel (net_forward_w2.hip) on the GPU box:
structured networks that localise an indexing mistake (centre-tap identity, single shifted taps, one channel
pair), the seeded synthetic networks against the CPU oracle and the reference-recorded fp64 logits, and the
forward-only rate of every 9x9 kernel.   python tools/check_w2.py [quick]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.helpers import load_npz
from oracle.net import OracleNet, make_state_dict
from tamago_amd.nn.network.dual_net import DualNet
from tamago_amd import lib as tl

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
dev = torch.device("cuda:0")
lib = tl.load()


def run(algo, sd, x, logits=True):
    os.environ["TG_FWD_ALGO"] = algo
    net = DualNet(dev, 9)
    net.load_state_dict(sd)
    out = net.inference_with_policy_logits(x) if logits else net.inference(x)
    name = lib.tg_net_kernel_name(net.handle, x.shape[0]).decode()
    return out, name


def structured(kind):
    """make_state_dict(9, 3, 1.4) with the 3x3 tower weights replaced by a structure."""
    sd = make_state_dict(9, 3, 1.4)
    for k in list(sd):
        if k.startswith("blocks.") and k.endswith(".weight") and sd[k].dim() == 4:
            w = torch.zeros_like(sd[k])
            if kind == "identity":                      # centre tap, cout == cin
                for c in range(64):
                    w[c, c, 1, 1] = 0.5
            elif kind.startswith("tap"):                # one off-centre tap, cout == cin
                t = int(kind[3:])
                for c in range(64):
                    w[c, c, t // 3, t % 3] = 0.5
            elif kind == "perm":                        # centre tap, cout = (5 cin + 3) % 64
                for c in range(64):
                    w[(5 * c + 3) % 64, c, 1, 1] = 0.5
            sd[k] = w
    return sd


rs = np.random.RandomState(5)
x300 = torch.from_numpy(rs.randint(-1, 2, size=(300, 6, 9, 9)).astype(np.float32))
print("== structured networks, B = 300 (w2 kernel vs CPU oracle; split16 for comparison) ==")
for kind in ["identity", "perm", "tap0", "tap5", "tap7", "random"]:
    sd = make_state_dict(9, 3, 1.4) if kind == "random" else structured(kind)
    ref = OracleNet(sd).inference_with_policy_logits(x300)
    for algo in ("w2", "split16"):
        (lg, val), name = run(algo, sd, x300)
        bad_boards = int(((lg - ref[0]).abs().amax(dim=1) > 1e-4).sum())
        print(f"{kind:9s} {algo:8s} {name:40s} logit err {float((lg - ref[0]).abs().max()):.3e}  value err "
              f"{float((val - ref[1]).abs().max()):.3e}  boards off: {bad_boards}/300", flush=True)

print("== accuracy against the reference-recorded fp64 forward (planes tiled to B > 256) ==")
fix = load_npz("net_s9.npz")
for algo in ("w2", "split16", "wino"):
    for seed in (0, 7):
        sd = make_state_dict(9, seed, float(fix[f"w{seed}_gain"]))
        x = torch.from_numpy(fix[f"w{seed}_planes"].astype(np.float32))
        n = x.shape[0]
        reps = (300 + n - 1) // n
        xx = x.repeat(reps, 1, 1, 1)
        (lg, val), name = run(algo, sd, xx)
        lg = lg.numpy().reshape(reps, n, -1)
        e64 = np.abs(lg - fix[f"w{seed}_logits64"][None]).max()
        eref = np.abs(fix[f"w{seed}_logits"] - fix[f"w{seed}_logits64"]).max()
        print(f"{algo:8s} seed {seed} B={xx.shape[0]} {name}: |logit - fp64| {e64:.3e} (reference fp32 path {eref:.3e})", flush=True)

print("== ragged batches vs oracle ==")
sd = make_state_dict(9, 7, 1.5)
ora = OracleNet(sd)
for b in (257, 770, 1539, 4099):
    x = torch.from_numpy(rs.randint(-1, 2, size=(b, 6, 9, 9)).astype(np.float32))
    rp, rv = ora.inference(x)
    (pol, val), name = run("w2", sd, x, logits=False)
    print(f"B={b:5d} {name}: policy err {float((pol - rp).abs().max()):.3e} value err {float((val - rv).abs().max()):.3e}", flush=True)

if not quick:
    print("== forward-only rate, planes resident ==")
    flops = lib.tg_net_flops_per_position(9)
    for algo in ("w2", "split16"):
        os.environ["TG_FWD_ALGO"] = algo
        torch.manual_seed(0)
        net = DualNet(dev, 9)
        for b in (768, 4096, 65536, 524288):
            x = torch.randint(-1, 2, (b, 6, 9, 9), device="cuda").float()
            out = (torch.empty((b, 82), device="cuda"), torch.empty((b, 3), device="cuda"))
            for _ in range(2):
                net.forward_device(x, out=out)
            torch.cuda.synchronize()
            iters = 10 if b < 100000 else 3
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                net.forward_device(x, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            print(f"{algo:8s} B={b:6d} {lib.tg_net_kernel_name(net.handle, b).decode():40s} {ms * 1e3:10.1f} us "
                  f"{b / ms * 1e3 / 1e6:7.3f} M pos/s  {b * flops / ms / 1e9:8.1f} TFLOP/s algorithmic", flush=True)
