"""Positions/s of the fp32 RL training step (tamago_amd/nn/learn.py) at the reference's batch
size (learning_param.py BATCH_SIZE = 256) and at larger batches, synthetic data in HBM."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tamago_amd.nn import learn  # noqa: E402
from oracle import train_ref  # noqa: E402   (torch-autograd comparison legs)

dev = torch.device("cuda", 0)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 9
batches = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256, 1024, 4096]
hip_only = len(sys.argv) > 3 and sys.argv[3] == "hip"     # profile runs: only this repo's kernels
for batch in batches:
    if hip_only:
        rng = np.random.RandomState(1)
        planes = torch.from_numpy((rng.uniform(size=(batch, 6, size, size)) < 0.3).astype(np.float32)).to(dev)
        pol = torch.softmax(torch.randn(batch, size * size + 1, device=dev), 1)
        val = torch.randint(0, 3, (batch,), device=dev)
        hip = learn.HipTrainer(dev, size, batch)
        for _ in range(3):
            hip.step(planes, pol, val)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(30):
            hip.step(planes, pol, val)
        torch.cuda.synchronize()
        dh = (time.time() - t0) / 30
        print(f"train step {size}x{size} batch {batch}: HIP kernels (tg_trainer_step) {dh * 1e3:.2f} ms -> {batch / dh:,.0f} positions/s")
        continue
    net = train_ref.TrainableDualNet(dev, size)
    opt = learn.make_optimizer(net, 0.01)
    rng = np.random.RandomState(1)
    planes = torch.from_numpy((rng.uniform(size=(batch, 6, size, size)) < 0.3).astype(np.float32)).to(dev)
    pol = torch.softmax(torch.randn(batch, size * size + 1, device=dev), 1)
    val = torch.randint(0, 3, (batch,), device=dev)
    for _ in range(5):
        train_ref.rl_train_step(net, opt, planes, pol, val)
    torch.cuda.synchronize()
    n = 30
    t0 = time.time()
    for _ in range(n):
        train_ref.rl_train_step(net, opt, planes, pol, val)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    run = train_ref.GraphedStep(net, opt, batch, "rl")
    for _ in range(3):
        run(planes, pol, val)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        run(planes, pol, val)
    torch.cuda.synchronize()
    dg = (time.time() - t0) / n
    # the reference's own GPU mode: the same step under fp16 autocast with a GradScaler (learn.py:342,371), eager as it runs it
    amp_net = train_ref.TrainableDualNet(dev, size)
    amp_opt = learn.make_optimizer(amp_net, 0.01)
    scaler = torch.amp.GradScaler("cuda")

    def amp_step():
        with torch.enable_grad():
            with torch.autocast(device_type="cuda", dtype=torch.float16):
                pp, vp = amp_net.forward(planes)
                loss = (learn.calculate_policy_kld_loss(pp, pol) + learn.RL_VALUE_WEIGHT * learn.calculate_value_loss(vp, val)).mean()
            amp_net.zero_grad()
            scaler.scale(loss).backward()
        scaler.step(amp_opt)
        scaler.update()
    for _ in range(5):
        amp_step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        amp_step()
    torch.cuda.synchronize()
    da = (time.time() - t0) / n
    line = (f"train step {size}x{size} batch {batch}: autograd eager {dt * 1e3:.2f} ms -> {batch / dt:,.0f} positions/s; "
            f"autograd hipGraph {dg * 1e3:.2f} ms -> {batch / dg:,.0f} positions/s; "
            f"autograd fp16 autocast + GradScaler, eager (the reference's GPU mode) {da * 1e3:.2f} ms")
    if size in (9, 19):
        hip = learn.HipTrainer(dev, size, batch)
        for _ in range(3):
            hip.step(planes, pol, val)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(n):
            hip.step(planes, pol, val)
        torch.cuda.synchronize()
        dh = (time.time() - t0) / n
        line += f"; HIP kernels (tg_trainer_step) {dh * 1e3:.2f} ms -> {batch / dh:,.0f} positions/s"
        hip.close()
    print(line, flush=True)
