#!/usr/bin/env python3
"""19x19 one-axis Winograd kernel over two workgroups per board (TG_FWD_ALGO=w1dband) on the GPU box: results against the CPU
oracle for a ladder of diagnostic networks (which part of the kernel a wrong answer comes from), then random networks at
several launch sizes, bit-identity across launch sizes, throughput next to the current 19x19 kernels.
    python tools/check_w1dband.py [--quick] [--diag]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle.net import OracleNet, make_state_dict
from tamago_amd.nn.network.dual_net import DualNet

S, P = 19, 361


def net_for(algo, sd):
    if algo:
        os.environ["TG_FWD_ALGO"] = algo
    else:
        os.environ.pop("TG_FWD_ALGO", None)
    net = DualNet(torch.device("cuda:0"), S)
    net.load_state_dict(sd)
    return net


def tower_keys(sd):
    return [k for k in sd if k.startswith("blocks.") and k.endswith("weight") and ".conv" in k]


def diag_nets():
    """(name, state dict): tower convolutions restricted to a subset of taps"""
    base = make_state_dict(S, 7, 1.5)
    out = []
    masks = {
        "tower weights zero (stem, residual path, heads)": np.zeros((3, 3)),
        "centre tap only (channel mixing)": np.array([[0, 0, 0], [0, 1, 0], [0, 0, 0]]),
        "horizontal taps only (x transform)": np.array([[0, 0, 0], [1, 1, 1], [0, 0, 0]]),
        "vertical taps only (y taps, halo hand-off)": np.array([[0, 1, 0], [0, 1, 0], [0, 1, 0]]),
        "upper row only (ky = 0)": np.array([[1, 1, 1], [0, 0, 0], [0, 0, 0]]),
        "lower row only (ky = 2)": np.array([[0, 0, 0], [0, 0, 0], [1, 1, 1]]),
        "all taps": np.ones((3, 3)),
    }
    for name, m in masks.items():
        sd = {k: v.clone() for k, v in base.items()}
        for k in tower_keys(sd):
            sd[k] = sd[k] * torch.from_numpy(m.astype(np.float32))[None, None]
        out.append((name, sd))
    return out


def compare(net, ora, x, tag):
    rp, rv = ora.inference(x)
    rl, _ = ora.inference_with_policy_logits(x)
    pol, val = net.inference(x)
    lg, _ = net.inference_with_policy_logits(x)
    ep, ev, el = float((pol - rp).abs().max()), float((val - rv).abs().max()), float((lg - rl).abs().max())
    bad = not (ep < 1e-4 and ev < 1e-4)
    msg = f"{tag:58s} B={x.shape[0]:5d} policy err {ep:.2e} value err {ev:.2e} logit err {el:.2e}"
    if bad:
        d = (lg - rl).abs()
        b, a = np.unravel_index(int(d.argmax()), d.shape)
        msg += f"   <-- FAIL (worst logit: board {b}, move {a} = (y {a // S}, x {a % S}))"
        per_board = d.amax(dim=1)
        msg += f"; boards over 1e-3: {[int(i) for i in torch.nonzero(per_board > 1e-3).flatten()[:12]]}"
    print(msg, flush=True)
    return not bad


def main():
    quick = "--quick" in sys.argv
    rs = np.random.RandomState(11)
    ok = True
    if "--diag" in sys.argv or not quick:
        for name, sd in diag_nets():
            ora = OracleNet(sd)
            net = net_for("w1dband", sd)
            for b in (1, 3):
                x = torch.from_numpy(rs.randint(-1, 2, size=(b, 6, S, S)).astype(np.float32))
                ok &= compare(net, ora, x, name)
            print("   range fallbacks:", net.range_fallbacks(), " band time-outs:", net.band_timeouts(), flush=True)
    sd = make_state_dict(S, 4, 1.2)
    ora = OracleNet(sd)
    net = net_for("w1dband", sd)
    for b in ((1, 2, 5, 64) if quick else (1, 2, 5, 64, 130, 300)):
        x = torch.from_numpy(rs.randint(-1, 2, size=(b, 6, S, S)).astype(np.float32))
        ok &= compare(net, ora, x, "random network, seed 4")
    print("   range fallbacks:", net.range_fallbacks(), " band time-outs:", net.band_timeouts(), flush=True)
    # bit-identity across launch sizes and repeated launches
    x = torch.from_numpy(rs.randint(-1, 2, size=(600, 6, S, S)).astype(np.float32))   # (above 512: the sixteen-boards-per-workgroup heads kernel)
    big = net.inference_with_policy_logits(x)
    again = net.inference_with_policy_logits(x)
    small = net.inference_with_policy_logits(x[:40])
    one = net.inference_with_policy_logits(x[7:8])
    same = torch.equal(big[0], again[0]) and torch.equal(big[0][:40], small[0]) and torch.equal(big[1][:40], small[1]) and \
        torch.equal(big[0][7:8], one[0]) and torch.equal(big[1][7:8], one[1])
    ok &= same
    print("600-position launch vs a repeat / 40 / 1 of the same positions:", "bit-identical" if same else "DIFFERENT   <-- FAIL", flush=True)
    # throughput, device-resident planes
    for b in ((64, 4096) if quick else (1, 64, 128, 256, 1024, 4096, 16384)):
        x = torch.from_numpy(rs.randint(-1, 2, size=(b, 6, S, S)).astype(np.float32)).cuda()
        for algo in ("w1dband", None):
            n2 = net_for(algo, sd)
            for _ in range(3):
                n2.forward_device(x)
            torch.cuda.synchronize()
            n = 10 if b >= 4096 else 100
            t0 = time.perf_counter()
            for _ in range(n):
                n2.forward_device(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            from tamago_amd import lib as tl
            name = tl.load().tg_net_kernel_name(n2.handle, b).decode()
            print(f"B={b:6d} {str(algo):8s} {dt * 1e6:9.1f} us  {b / dt / 1e6:6.3f} M positions/s   ({name}; fallbacks {n2.range_fallbacks()})", flush=True)
    print("OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
