"""Phase timings inside the training-step kernels (tools/experiments/train_prof.sh builds the stamped library).
Prints, for workgroup 0 of four launches of a step, the time between consecutive stamps (100 MHz clock -> 10 ns steps)."""
import os
import sys

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import tamago_amd.lib as tl  # noqa: E402
tl.LIB_PATH = os.path.join(ROOT, "tools/experiments/_bin/libtamago_trainprof.so")
from tamago_amd.nn import learn  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
rng = np.random.RandomState(1)
planes = torch.from_numpy((rng.uniform(size=(batch, 6, 9, 9)) < 0.3).astype(np.float32)).to(dev)
pol = torch.softmax(torch.randn(batch, 82, device=dev), 1)
val = torch.randint(0, 3, (batch,), device=dev)
hip = learn.HipTrainer(dev, 9, batch)
names = {0: ("conv FWD l=5", ["start", "tables", "staged", "mfma loop", "epilogue", "sums+atomics"]),
         1: ("conv DGRAD l=5", ["start", "tables", "staged", "mfma loop", "epilogue", "sums+atomics"]),
         2: ("wgrad l=5", ["start", "zero+tables+request", "board 0 deposited", "board 1", "board 2", "board 3", "last mfma loop", "partial written"]),
         3: ("head_loss", ["start", "tables", "h staged", "fc forward", "softmax", "fc backward", "reduce+atomics"])}
acc = {}
lib = tl.load()
for it in range(12):
    hip.step(planes, pol, val)
    out = np.zeros(64, np.uint64)
    tl.check(lib.tg_trainer_debug_read(hip.handle, 3, 0, out.ctypes.data))
    if it >= 2:
        for slot, (nm, labels) in names.items():
            st = out[slot * 16: slot * 16 + len(labels)].astype(np.int64)
            acc.setdefault(slot, []).append(np.diff(st) * 10)
for slot, (nm, labels) in names.items():
    d = np.mean(acc[slot], 0)
    print(f"{nm}: total {d.sum() / 1000:.2f} us | " + ", ".join(f"{lab} {v / 1000:.2f}" for lab, v in zip(labels[1:], d)))
