#!/usr/bin/env python3
"""One-axis Winograd forward kernel (TG_FWD_ALGO=w1d, the 9x9 default) on the GPU box: results against the CPU oracle at every
workgroup shape, accuracy against the reference-recorded fp64 forward, throughput next to the direct split kernel.
    python tools/check_w1d.py [--quick]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle.net import OracleNet, make_state_dict
from tamago_amd.nn.network.dual_net import DualNet
from tests.helpers import load_npz


ALGOS = ("w1d", "split16")


def net_for(algo, sd):
    os.environ["TG_FWD_ALGO"] = algo
    net = DualNet(torch.device("cuda:0"), 9)
    net.load_state_dict(sd)
    return net


def main():
    quick = "--quick" in sys.argv
    sd = make_state_dict(9, 7, 1.5)
    ora = OracleNet(sd)
    rs = np.random.RandomState(11)
    ok = True
    for b in (1, 5, 256, 300, 512, 769, 1301, 4096):
        x = torch.from_numpy(rs.randint(-1, 2, size=(b, 6, 9, 9)).astype(np.float32))
        rp, rv = ora.inference(x)
        rl, _ = ora.inference_with_policy_logits(x)
        for algo in ALGOS:
            net = net_for(algo, sd)
            pol, val = net.inference(x)
            lg, _ = net.inference_with_policy_logits(x)
            ep, ev = float((pol - rp).abs().max()), float((val - rv).abs().max())
            el = float((lg - rl).abs().max())
            bad = not (ep < 1e-4 and ev < 1e-4)
            ok &= not bad
            print(f"B={b:5d} {algo:8s} policy err {ep:.2e} value err {ev:.2e} logit err {el:.2e} (max |logit| {float(rl.abs().max()):.1f})"
                  + ("   <-- FAIL" if bad else ""), flush=True)
            if bad and algo == "w1d":
                d = (pol - rp).abs().amax(dim=1)
                print("   boards over tolerance:", [int(i) for i in torch.nonzero(d > 1e-4).flatten()[:20]], "of", b)
    # results must not depend on the launch size: the one-board and the three-board variants of a kernel family, bit for bit
    x = torch.from_numpy(rs.randint(-1, 2, size=(600, 6, 9, 9)).astype(np.float32))
    for algo in ALGOS:
        net = net_for(algo, sd)
        big = net.inference_with_policy_logits(x)
        small = net.inference_with_policy_logits(x[:100])
        one = net.inference_with_policy_logits(x[7:8])
        same = torch.equal(big[0][:100], small[0]) and torch.equal(big[1][:100], small[1]) and torch.equal(big[0][7:8], one[0]) and torch.equal(big[1][7:8], one[1])
        ok &= same
        print(f"{algo:8s} 600-position launch vs 100 / 1 of the same positions: {'bit-identical' if same else 'DIFFERENT   <-- FAIL'}", flush=True)
    fix = load_npz("net_s9.npz")
    for seed in (0, 7):
        sdf = make_state_dict(9, seed, float(fix[f"w{seed}_gain"]))
        x = torch.from_numpy(fix[f"w{seed}_planes"].astype(np.float32))
        n = x.shape[0]
        for reps in (1, (300 + n - 1) // n):
            for algo in ALGOS:
                lg, _ = net_for(algo, sdf).inference_with_policy_logits(x.repeat(reps, 1, 1, 1))
                e64 = np.abs(lg[:n].numpy() - fix[f"w{seed}_logits64"]).max()
                eref = np.abs(fix[f"w{seed}_logits"] - fix[f"w{seed}_logits64"]).max()
                print(f"seed {seed} B={n * reps:4d} {algo:8s} |logit - fp64| {e64:.3e}  reference fp32 path {eref:.3e}  ratio {e64 / eref:.2f}"
                      + ("   <-- over 4x" if e64 >= 4 * eref + 1e-6 else ""), flush=True)
    # throughput, device-resident planes
    for b in ((65536,) if quick else (256, 768, 1600, 65536, 196608)):
        x = torch.from_numpy(rs.randint(-1, 2, size=(b, 6, 9, 9)).astype(np.float32)).cuda()
        for algo in ALGOS:
            net = net_for(algo, sd)
            for _ in range(3):
                net.forward_device(x)
            torch.cuda.synchronize()
            n = 20 if b >= 65536 else 200
            t0 = time.perf_counter()
            for _ in range(n):
                net.forward_device(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f"B={b:6d} {algo:8s} {dt * 1e6:9.1f} us  {b / dt / 1e6:6.3f} M positions/s", flush=True)
    print("OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
