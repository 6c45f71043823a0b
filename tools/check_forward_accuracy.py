#!/usr/bin/env python3
"""Accuracy of every 9x9 forward kernel against the reference-recorded fp64 forward (tests/golden/net_s9.npz)
and against the CPU oracle on random planes.  GPU box: python tools/check_forward_accuracy.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import load_npz
from oracle.net import OracleNet, make_state_dict
from tamago_amd.nn.network.dual_net import DualNet
fix = load_npz("net_s9.npz")
for algo in ("wino", "direct", "split16", "w1d"):
    os.environ["TG_FWD_ALGO"] = algo
    for seed in (0, 7):
        sd = make_state_dict(9, seed, float(fix[f"w{seed}_gain"]))
        net = DualNet(torch.device("cuda:0"), 9); net.load_state_dict(sd)
        x = torch.from_numpy(fix[f"w{seed}_planes"].astype(np.float32))
        for reps in (1, (300 + x.shape[0] - 1) // x.shape[0]):       # as recorded (one board per workgroup) and tiled
            xx = x.repeat(reps, 1, 1, 1)                              # beyond the CU count (three boards per workgroup)
            lg, val = net.inference_with_policy_logits(xx)
            pol, _ = net.inference(xx)
            lg, val, pol = lg[:x.shape[0]], val[:x.shape[0]], pol[:x.shape[0]]
            if reps > 1:
                print(f"{algo:10s} seed {seed} B={xx.shape[0]}: |logit-fp64| {np.abs(lg.numpy() - fix[f'w{seed}_logits64']).max():.3e}")
        e64 = np.abs(lg.numpy() - fix[f"w{seed}_logits64"]).max()
        eref = np.abs(fix[f"w{seed}_logits"] - fix[f"w{seed}_logits64"]).max()
        ep = np.abs(pol.numpy() - fix[f"w{seed}_policy"]).max()
        ev = np.abs(val.numpy() - fix[f"w{seed}_value"]).max()
        print(f"{algo:10s} seed {seed} B={x.shape[0]}: |logit-fp64| {e64:.3e} (reference fp32 path {eref:.3e}, max|logit| {np.abs(fix[f'w{seed}_logits64']).max():.1f}); policy err {ep:.2e} value err {ev:.2e}")
    # random planes vs oracle fp32 and an fp64 oracle
    sd = make_state_dict(9, 7, 1.5)
    net = DualNet(torch.device("cuda:0"), 9); net.load_state_dict(sd)
    ora = OracleNet(sd)
    sd64 = {k: v.double() for k, v in sd.items()}
    rs = np.random.RandomState(11)
    for b in (5, 300):
        x = torch.from_numpy(rs.randint(-1, 2, size=(b, 6, 9, 9)).astype(np.float32))
        lg, val = net.inference_with_policy_logits(x)
        rl, rv = ora.inference_with_policy_logits(x)
        print(f"{algo:10s} random B={b}: logit err vs oracle fp32 {float((lg-rl).abs().max()):.3e} (max |logit| {float(rl.abs().max()):.1f}), value err {float((val-rv).abs().max()):.3e}")
