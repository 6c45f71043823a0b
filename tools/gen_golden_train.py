"""Golden vectors for the training step (tests/test_gpu_train.py), produced by the REFERENCE
modules on the CPU in fp32: nn/network/dual_net.py + nn/loss.py + torch.optim.SGD as
nn/learn.py:333-376 wires them (without autocast - the CPU has no fp16 path).

    python tools/gen_golden_train.py        # needs /root/reference; writes tests/golden/train_s9.npz and train_s19.npz

The reference's DualNet takes the board size as a constructor argument (nn/network/dual_net.py:17), so 19x19 needs no
scratch copy of the reference here (batch 16 at 19x19: the fixture stays small, the step is the same code).

Inputs are rebuilt from seeds by `make_case` (numpy RandomState, portable), so the fixture holds
outputs only: per-step losses, a 48-value sample of every parameter after every step, and the
full batch-norm statistics.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tamago_amd.nn.network.dual_net import state_dict_keys  # noqa: E402

STEPS, BATCH, SIZE = 3, 32, 9
BATCH_BY_SIZE = {9: BATCH, 19: 16}


def make_case(seed=20240, SIZE=SIZE):
    rng = np.random.RandomState(seed + (0 if SIZE == 9 else SIZE))
    BATCH = BATCH_BY_SIZE[SIZE]
    state = {}
    for key, shape in state_dict_keys(SIZE):
        if key.endswith("running_mean"):
            v = rng.normal(0, 0.05, shape)
        elif key.endswith("running_var"):
            v = rng.uniform(0.8, 1.2, shape)
        elif ".bn" in key or key.startswith("bn_layer"):
            v = rng.uniform(0.7, 1.3, shape) if key.endswith("weight") else rng.normal(0, 0.1, shape)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 64
            v = rng.uniform(-1, 1, shape) / np.sqrt(fan_in)
        state[key] = torch.from_numpy(np.asarray(v, np.float32))
    batches = []
    for _ in range(STEPS):
        # continuous planes, not 0/1 boards: on binary planes whole regions of a channel share
        # one pre-activation value, and when that value rounds to either side of zero the ReLU
        # mask of the region flips as a block - fp32 gradients then differ from fp64 ones by
        # percents (measured: 2.2 % on blocks.4.conv1.weight, CPU fp32 vs CPU fp64), which
        # says nothing about the step's arithmetic.  Continuous inputs have no such ties.
        planes = rng.uniform(size=(BATCH, 6, SIZE, SIZE)).astype(np.float32)
        pol = rng.gamma(0.3, size=(BATCH, SIZE * SIZE + 1)).astype(np.float64)
        pol = (pol / pol.sum(1, keepdims=True)).astype(np.float32)
        val = rng.randint(0, 3, BATCH).astype(np.int64)
        batches.append((planes, pol, val))
    return state, batches


def sample_of(t):
    flat = t.reshape(-1)
    idx = np.linspace(0, flat.size - 1, min(48, flat.size)).astype(np.int64)
    return flat[idx]


def main():
    for size in (9, 19):
        generate(size)


def generate(SIZE):
    sys.path.insert(0, "/root/reference")
    from nn.network.dual_net import DualNet
    from nn.loss import calculate_policy_kld_loss, calculate_value_loss, calculate_policy_loss
    torch.set_num_threads(4)
    out = {}
    for mode in ("rl", "sl"):
        state, batches = make_case(SIZE=SIZE)
        net = DualNet(torch.device("cpu"), SIZE)
        full = dict(net.state_dict())
        full.update(state)
        net.load_state_dict(full)
        net.train()
        opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4,
                              nesterov=True)
        losses = []
        for planes, pol, val in batches:
            x, p, v = torch.tensor(planes), torch.tensor(pol), torch.tensor(val)
            if mode == "rl":
                pp, vp = net.forward(x)
                net.zero_grad()
                pl = calculate_policy_kld_loss(pp, p)
                vl = calculate_value_loss(vp, v)
                loss = (pl + 1.0 * vl).mean()
            else:
                pp, vp = net.forward_for_sl(x)
                net.zero_grad()
                pl = calculate_policy_loss(pp, p)
                vl = calculate_value_loss(vp, v)
                loss = (pl + 0.02 * vl).mean()
            loss.backward()
            opt.step()
            losses.append([loss.item(), pl.mean().item(), vl.mean().item()])
            snap = net.state_dict()
            for key, _ in state_dict_keys(SIZE):
                if not key.endswith(("running_mean", "running_var")):
                    out[f"{mode}/step{len(losses)}/{key}"] = sample_of(snap[key].detach().numpy()).copy()
        out[f"{mode}_losses"] = np.asarray(losses, np.float64)
        final = net.state_dict()
        for key, _ in state_dict_keys(SIZE):
            arr = final[key].detach().numpy()
            if key.endswith(("running_mean", "running_var")):
                out[f"{mode}/{key}"] = arr.copy()
        net.eval()
        with torch.no_grad():
            pe, ve = net.forward(torch.tensor(batches[0][0]))
        out[f"{mode}_eval_policy"] = pe.numpy()
        out[f"{mode}_eval_value"] = ve.numpy()
    path = os.path.join(ROOT, "tests", "golden", f"train_s{SIZE}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes", out["rl_losses"], out["sl_losses"])


if __name__ == "__main__":
    main()
