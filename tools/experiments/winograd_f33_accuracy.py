# Numerical feasibility of Winograd F(3x3,3x3) (5x5 input tiles, 9x9 = 3x3 tiles exactly) in float32
# for the DualNet tower: error vs a float64 direct convolution through 12 layers with BN + residual + ReLU.
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.net import make_state_dict
# F(3,3) with points 0, 1, -1, 2, inf (Lavin & Gray style construction via Cook-Toom)
def cook_toom(m, r, pts):
    # returns AT (m x a), G (a x r), BT (a x a) with a = m + r - 1, last point = infinity
    a = m + r - 1
    from fractions import Fraction as F
    pts = [F(p) for p in pts]           # a-1 finite points
    # Lagrange basis
    import itertools
    def poly_mul(p, q):
        out = [F(0)] * (len(p) + len(q) - 1)
        for i, x in enumerate(p):
            for j, y in enumerate(q): out[i + j] += x * y
        return out
    M = [F(1)]
    for p in pts: M = poly_mul(M, [-p, F(1)])        # M(x) = prod (x - p_i), degree a-1
    AT = [[(p ** i) for p in pts] + [F(1) if i == m - 1 else F(0)] for i in range(m)]
    G = []
    for k, p in enumerate(pts):
        denom = F(1)
        for j, q in enumerate(pts):
            if j != k: denom *= (p - q)
        G.append([(p ** i) / denom for i in range(r)])
    G.append([F(0)] * (r - 1) + [F(1)])
    BT = []
    for k, p in enumerate(pts):
        # M(x)/(x - p_k) coefficients
        num = [F(1)]
        for j, q in enumerate(pts):
            if j != k: num = poly_mul(num, [-q, F(1)])
        BT.append(num + [F(0)] * (a - len(num)))
    BT.append(M[:a])
    f = lambda mat: np.array([[float(x) for x in row] for row in mat], dtype=np.float64)
    return f(AT), f(G), f(BT)
AT, G, BT = cook_toom(3, 3, [0, 1, -1, 2])
# sanity 1-D
d = np.random.randn(5); g = np.random.randn(3)
y = AT @ ((G @ g) * (BT @ d)); ref = np.array([d[i:i+3] @ g for i in range(3)])
print("1-D check", np.abs(y - ref).max())
def conv_ref(x, w):   # x [C,9,9] f64, w [O,C,3,3]
    return torch.nn.functional.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), padding=1)[0].numpy()
def conv_wino(x, w, dt):
    C = x.shape[0]; O = w.shape[0]
    U = np.einsum("ai,ocij,bj->ocab", G, w.astype(np.float64), G).astype(dt)      # host transform in f64, stored f32
    xp = np.zeros((C, 11, 11), dtype=dt); xp[:, 1:10, 1:10] = x
    out = np.zeros((O, 9, 9), dtype=dt)
    BTd, ATd = BT.astype(dt), AT.astype(dt)
    for ty in range(3):
        for tx in range(3):
            patch = xp[:, 3*ty:3*ty+5, 3*tx:3*tx+5]
            V = np.einsum("ai,cij,bj->cab", BTd, patch, BTd).astype(dt)
            Mm = np.einsum("ocab,cab->oab", U, V).astype(dt)
            Y = np.einsum("ia,oab,jb->oij", ATd, Mm, ATd).astype(dt)
            out[:, 3*ty:3*ty+3, 3*tx:3*tx+3] = Y
    return out
sd = make_state_dict(9, 7, 1.5)
rs = np.random.RandomState(0)
x64 = rs.randint(-1, 2, size=(6, 9, 9)).astype(np.float64)
def bn(prefix, eps):
    w, b, m, v = (sd[f"{prefix}.{k}"].double().numpy() for k in ("weight", "bias", "running_mean", "running_var"))
    s = w / np.sqrt(v + eps); return s, b - m * s
def run(dt, wino):
    x = conv_ref(x64, sd["conv_layer.weight"].double().numpy())
    s, t = bn("bn_layer", 1e-5); x = np.maximum(x * s[:, None, None] + t[:, None, None], 0).astype(dt)
    for b in range(6):
        res = x
        for c in (1, 2):
            w = sd[f"blocks.{b}.conv{c}.weight"].double().numpy()
            y = conv_wino(x, w, dt) if wino else conv_ref(x.astype(np.float64), w).astype(dt)
            s, t = bn(f"blocks.{b}.bn{c}", 2e-5)
            y = (y * s[:, None, None].astype(dt) + t[:, None, None].astype(dt)).astype(dt)
            if c == 2: y = y + res
            x = np.maximum(y, 0).astype(dt)
    return x
ref = run(np.float64, False)
for name, dt, wino in (("direct f32", np.float32, False), ("F(3,3) f32", np.float32, True), ("F(3,3) f64", np.float64, True)):
    got = run(dt, wino)
    print(f"{name}: max abs err after 12 layers {np.abs(got - ref).max():.3e}, activation max {np.abs(ref).max():.2f}")
