# Numerical feasibility of Winograd F(2x2,3x3) on f16 x 2 SPLIT operands (round 4): emulates on the CPU what
# dualnet_fwd_wsplit_kernel computes - V = B^T d B in fp32, split into two f16 pieces; U = G g G^T in fp64, scaled by
# a power of two per layer, split into two f16 pieces; M = Vh Uh + 2^-11 (Vh Ul + Vl Uh) with fp32 accumulation;
# Y = A^T M A in fp32; folded batch norm, residual, ReLU - and compares the logits with the reference-recorded fp64
# forward (tests/golden/net_s9.npz), next to the reference's own fp32 path.  Test criterion: err <= 4 x err_ref + 1e-6.
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.net import make_state_dict, EPS_STEM, EPS_BODY   # noqa: E402
from tests.helpers import load_npz                            # noqa: E402

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def split16(x32):
    """x (fp32 array) -> (hi, lo') as float32 arrays holding f16 values, x ~ hi + lo' / 2048"""
    hi = x32.astype(np.float16)
    lo = ((x32 - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def fold(sd, prefix, eps):
    w, b, m, v = (sd[f"{prefix}.{k}"].double().numpy() for k in ("weight", "bias", "running_mean", "running_var"))
    s = w / np.sqrt(v + eps)
    return s.astype(np.float32), (b - m * s).astype(np.float32)


def conv_wino_split(x, w, mode):
    """x [B,64,9,9] fp32, w [64,64,3,3] fp32 -> [B,64,9,9] fp32 (no BN).  mode: 'split' | 'fp32' """
    B = x.shape[0]
    U = np.einsum("ai,ocij,bj->abco", G, w.astype(np.float64), G)            # [4,4,cin,cout] fp64
    xp = np.zeros((B, 64, 12, 12), dtype=np.float32)
    xp[:, :, 1:10, 1:10] = x
    # tiles: 5 x 5 per board, patch (ty,tx) = xp[2ty : 2ty+4, 2tx : 2tx+4]
    idx = (2 * np.arange(5))[:, None] + np.arange(4)[None, :]                # [5,4]
    patch = xp[:, :, idx[:, None, :, None], idx[None, :, None, :]]           # [B,64,5,5,4,4]
    bt = BT.astype(np.float32)
    # fp32 adds, in the kernel's order: rows first, then columns (entries are 0, +-1: exact products)
    t = np.einsum("ai,bcyxij->bcyxaj", bt, patch).astype(np.float32)
    V = np.einsum("bcyxaj,dj->bcyxad", t, bt).astype(np.float32)            # [B,cin,5,5,4,4]
    if mode == "fp32":
        M = np.einsum("bcyxad,adco->boyxad", V.astype(np.float64), U.astype(np.float32).astype(np.float64)).astype(np.float32)
    else:
        mx = np.abs(U).max()
        e = 10 - int(np.frexp(mx)[1])
        Us = (U * 2.0 ** e).astype(np.float32)
        Uh, Ul = split16(Us)
        Vh, Vl = split16(V)
        tV = lambda a: torch.from_numpy(np.ascontiguousarray(a.transpose(4, 5, 0, 2, 3, 1).reshape(16, -1, 64)))   # [pt][B*25][cin]
        tU = lambda a: torch.from_numpy(np.ascontiguousarray(a.reshape(16, 64, 64)))                               # [pt][cin][cout]
        main = torch.bmm(tV(Vh), tU(Uh))                                     # fp32 accumulate
        cross = torch.bmm(tV(Vh), tU(Ul)) + torch.bmm(tV(Vl), tU(Uh))
        Mm = (main + cross * np.float32(1.0 / 2048.0)) * np.float32(2.0 ** -e)
        M = Mm.numpy().reshape(4, 4, B, 5, 5, 64).transpose(2, 5, 3, 4, 0, 1)                                       # [B,cout,5,5,4,4]
    at = AT.astype(np.float32)
    t = np.einsum("ia,boyxad->boyxid", at, M).astype(np.float32)
    Y = np.einsum("boyxid,jd->boyxij", t, at).astype(np.float32)             # [B,cout,5,5,2,2]
    out = Y.transpose(0, 1, 2, 4, 3, 5).reshape(B, 64, 10, 10)[:, :, :9, :9]
    return np.ascontiguousarray(out)


def forward(sd, planes, mode):
    F = torch.nn.functional
    x = F.conv2d(torch.from_numpy(planes), sd["conv_layer.weight"], padding=1).numpy()
    s, t = fold(sd, "bn_layer", EPS_STEM)
    x = np.maximum(x * s[None, :, None, None] + t[None, :, None, None], 0).astype(np.float32)
    for b in range(6):
        res = x
        for c in (1, 2):
            y = conv_wino_split(x, sd[f"blocks.{b}.conv{c}.weight"].numpy(), mode)
            s, t = fold(sd, f"blocks.{b}.bn{c}", EPS_BODY)
            y = (y * s[None, :, None, None] + t[None, :, None, None]).astype(np.float32)
            if c == 2:
                y = y + res
            x = np.maximum(y, 0).astype(np.float32)
    xt = torch.from_numpy(x)
    bsz = x.shape[0]

    def bn(h, prefix):
        return F.batch_norm(h, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                            sd[prefix + ".bias"], False, 0.0, EPS_BODY)
    ph = F.relu(bn(F.conv2d(xt, sd["policy_head.conv_layer.weight"]), "policy_head.bn_layer"))
    return F.linear(ph.reshape(bsz, -1), sd["policy_head.fc_layer.weight"], sd["policy_head.fc_layer.bias"]).numpy()


if __name__ == "__main__":
    fix = load_npz("net_s9.npz")
    for seed in (0, 7):
        sd = make_state_dict(9, seed, float(fix[f"w{seed}_gain"]))
        x = fix[f"w{seed}_planes"].astype(np.float32)
        ref64 = fix[f"w{seed}_logits64"]
        eref = np.abs(fix[f"w{seed}_logits"] - ref64).max()
        for mode in ("fp32", "split"):
            lg = forward(sd, x, mode)
            e = np.abs(lg - ref64).max()
            print(f"seed {seed} winograd {mode:5s}: |logit - fp64| {e:.3e}   reference fp32 path {eref:.3e}   ratio {e / eref:.2f}"
                  f"   (criterion < {4 * eref + 1e-6:.3e})")
