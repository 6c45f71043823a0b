"""torch-autograd fp32 reference of the DualNet training step - TEST INFRASTRUCTURE, like the rest of oracle/.

The product's mini-batch step is `tamago_amd.nn.learn.HipTrainer` (tamago_amd/csrc/train.hip).  This module restates the
reference's step (nn/learn.py:318-403, 126-232; nn/loss.py:9-55; dual_net.py:41-52, res_block.py:27-40) with torch ops
(ATen / MIOpen on the device, the reference's own kernels on the CPU) and is what the HIP kernels are compared with
(tests/test_train_step.py, tools/bench_train.py).  It is pinned against vectors produced by the reference's modules
(tools/gen_golden_train.py -> tests/golden/train_s9.npz).  Nothing under tamago_amd/ imports it.
"""
import glob
import os
import sys
import time
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from tamago_amd.nn.learn import (BLOCKS, LEARNING_SCHEDULE, ParamTable, RL_VALUE_WEIGHT, SL_LEARNING_RATE, SL_VALUE_WEIGHT,
                                 _BODY_BN, _STEM_BN, _chunk_on_device, calculate_policy_kld_loss, calculate_policy_loss,
                                 calculate_value_loss, make_optimizer, print_learning_process, split_train_test_set)


class TrainableDualNet(ParamTable):
    """The table as leaves of an autograd graph + the forward pass in torch ops."""

    def train(self):
        self.training = True
        return self

    def eval(self):
        self.training = False
        return self

    def zero_grad(self):
        for p in self.parameters():
            p.grad = None

    # ---- forward (dual_net.py:41-52, res_block.py:27-40, head/*.py) -------------------------
    def _bn(self, x, prefix, cfg):
        eps, momentum = cfg
        t = self.t
        return F.batch_norm(x, t[prefix + ".running_mean"], t[prefix + ".running_var"],
                            t[prefix + ".weight"], t[prefix + ".bias"],
                            training=self.training, momentum=momentum, eps=eps)

    def forward(self, planes: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Policy logits [B, S*S+1] and value logits [B, 3]."""
        t = self.t
        self.batches_tracked += int(self.training)
        x = F.relu(self._bn(F.conv2d(planes, t["conv_layer.weight"], padding=1),
                            "bn_layer", _STEM_BN))
        for b in range(BLOCKS):
            pre = f"blocks.{b}"
            h = F.relu(self._bn(F.conv2d(x, t[pre + ".conv1.weight"], padding=1),
                                pre + ".bn1", _BODY_BN))
            h = self._bn(F.conv2d(h, t[pre + ".conv2.weight"], padding=1), pre + ".bn2", _BODY_BN)
            x = F.relu(x + h)
        heads = []
        for name in ("policy_head", "value_head"):
            h = F.relu(self._bn(F.conv2d(x, t[name + ".conv_layer.weight"]),
                                name + ".bn_layer", _BODY_BN))
            heads.append(F.linear(h.flatten(1), t[name + ".fc_layer.weight"],
                                  t[name + ".fc_layer.bias"]))
        return heads[0], heads[1]



def rl_train_step(net: TrainableDualNet, optimizer, plane, policy, value) -> Dict[str, float]:
    """One mini-batch of learn.py:360-376 (KLD policy loss + value cross entropy)."""
    with torch.enable_grad():
        policy_predict, value_predict = net.forward(plane)
        net.zero_grad()
        policy_loss = calculate_policy_kld_loss(policy_predict, policy)
        value_loss = calculate_value_loss(value_predict, value)
        loss = (policy_loss + RL_VALUE_WEIGHT * value_loss).mean()
        loss.backward()
    optimizer.step()
    return {"loss": loss.item(), "policy": policy_loss.mean().item(),
            "value": value_loss.mean().item()}


def sl_train_step(net: TrainableDualNet, optimizer, plane, policy, value) -> Dict[str, float]:
    """One mini-batch of the supervised trainer (learn.py:150-180): the policy target is a
    distribution scored against the softmax output, value weight 0.02."""
    with torch.enable_grad():
        policy_predict, value_predict = net.forward(plane)
        net.zero_grad()
        policy_loss = calculate_policy_loss(F.softmax(policy_predict, dim=1), policy)
        value_loss = calculate_value_loss(value_predict, value)
        loss = (policy_loss + SL_VALUE_WEIGHT * value_loss).mean()
        loss.backward()
    optimizer.step()
    return {"loss": loss.item(), "policy": policy_loss.mean().item(),
            "value": value_loss.mean().item()}


class GraphedStep:
    """One mini-batch step captured in a hipGraph (torch.cuda.CUDAGraph) and replayed: the
    eager step is launch-bound (a few hundred small kernels for 2.7 ms of a 256-position
    batch), a replay is one submission.  Inputs are copied into static buffers, the three
    loss values are accumulated on the device (no host read per step).  The operator sequence
    is the eager step's; results agree with it to summation-order noise."""

    def __init__(self, net: TrainableDualNet, optimizer, batch_size: int, mode: str = "rl"):
        dev, s = net.device, net.board_size
        self.net, self.optimizer = net, optimizer
        self.plane = torch.zeros((batch_size, 6, s, s), device=dev)
        self.policy = torch.full((batch_size, s * s + 1), 1.0 / (s * s + 1), device=dev)
        self.value = torch.zeros((batch_size,), dtype=torch.int64, device=dev)
        self.sums = torch.zeros(3, dtype=torch.float64, device=dev)
        self.steps = 0
        body = self._rl if mode == "rl" else self._sl
        # warm-up on a side stream (library workspaces, algorithm choice), then put every
        # tensor the warm-up touched back: parameters, statistics, momentum buffers
        params = [p.detach().clone() for p in net.parameters()]
        stats = {k: v.clone() for k, v in net.t.items() if not v.requires_grad}
        had = {id(p): optimizer.state[p]["momentum_buffer"].clone()
               for p in net.parameters() if "momentum_buffer" in optimizer.state.get(p, {})}
        tracked = net.batches_tracked
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.enable_grad():
            for _ in range(3):
                optimizer.zero_grad(set_to_none=True)
                body()
        torch.cuda.current_stream(dev).wait_stream(side)
        self._restore(params, stats, had)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.enable_grad(), torch.cuda.graph(self.graph):
            body()
        self._restore(params, stats, had)     # the capture pass does not execute, but be exact
        self.sums.zero_()
        net.batches_tracked = tracked

    def _restore(self, params, stats, had):
        with torch.no_grad():
            for p, saved in zip(self.net.parameters(), params):
                p.copy_(saved)
                buf = self.optimizer.state[p].get("momentum_buffer")
                if buf is not None:           # zeros == "no buffer yet": first step sets buf = grad
                    buf.copy_(had[id(p)]) if id(p) in had else buf.zero_()
            for k, saved in stats.items():
                self.net.t[k].copy_(saved)

    def _losses(self, policy_loss, value_loss, weight):
        loss = (policy_loss + weight * value_loss).mean()
        loss.backward()
        self.optimizer.step()
        self.sums += torch.stack([loss.detach(), policy_loss.detach().mean(),
                                  value_loss.detach().mean()]).double()

    def _rl(self):
        p, v = self.net.forward(self.plane)
        self._losses(calculate_policy_kld_loss(p, self.policy),
                     calculate_value_loss(v, self.value), RL_VALUE_WEIGHT)

    def _sl(self):
        p, v = self.net.forward(self.plane)
        self._losses(calculate_policy_loss(F.softmax(p, dim=1), self.policy),
                     calculate_value_loss(v, self.value), SL_VALUE_WEIGHT)

    def __call__(self, plane, policy, value):
        self.plane.copy_(plane, non_blocking=True)
        self.policy.copy_(policy, non_blocking=True)
        self.value.copy_(value, non_blocking=True)
        self.graph.replay()
        self.steps += 1
        self.net.batches_tracked += 1

    def take_losses(self) -> Dict[str, float]:
        """Summed losses since the last call (one host read)."""
        total = self.sums.tolist()
        self.sums.zero_()
        return {"loss": total[0], "policy": total[1], "value": total[2]}


def train_on_gpu_reference(program_dir: str, board_size: int, batch_size: int, epochs: int,
                 device_index: int = 0) -> Dict[str, float]:
    """torch-autograd restatement of the supervised trainer, learn.py:126-232 (the checker of
    tamago_amd.nn.learn.train_on_gpu): `epochs` passes over the training chunks of
    ``data/sl_data_*.npz``, after each one the test chunks in eval mode, then the learning
    rate schedule; writes ``model/sl-model.bin`` relative to the working directory, as the
    reference does (learn.py:232).  Returns the last test-loss sums."""
    if not torch.cuda.is_available():
        raise RuntimeError("tamago_amd trains on the GPU only")
    device = torch.device("cuda", device_index)
    torch.cuda.set_device(device)             # graph capture and side streams run on the CURRENT device
    data_set = sorted(glob.glob(os.path.join(program_dir, "data", "sl_data_*.npz")))
    train_files, test_files = split_train_test_set(data_set, 0.8)
    net = TrainableDualNet(device, board_size)
    optimizer = make_optimizer(net, SL_LEARNING_RATE)
    current_lr = SL_LEARNING_RATE
    eager = os.environ.get("TG_TRAIN_EAGER", "0") == "1"
    graphed = None
    test_loss = {"loss": 0.0, "policy": 0.0, "value": 0.0}
    for epoch in range(epochs):
        for data_index, path in enumerate(train_files):
            planes, policies, values = _chunk_on_device(path, device)
            net.train()
            if graphed is None and not eager:     # (re)captured after a learning-rate change:
                graphed = GraphedStep(net, optimizer, batch_size, "sl")   # lr is baked into the graph
            train_loss = {"loss": 0.0, "policy": 0.0, "value": 0.0}
            iteration = 0
            started = time.time()
            for i in range(0, len(values) - batch_size + 1, batch_size):
                batch = (planes[i:i + batch_size], policies[i:i + batch_size], values[i:i + batch_size])
                if eager:
                    part = sl_train_step(net, optimizer, *batch)
                    for k in train_loss:
                        train_loss[k] += part[k]
                else:
                    graphed(*batch)
                iteration += 1
            if not eager:
                train_loss = graphed.take_losses()
            print_learning_process(train_loss, epoch, data_index, iteration, started)

        sums = torch.zeros(3, dtype=torch.float64, device=device)
        test_iteration = 0
        started = time.time()
        net.eval()
        for path in test_files:
            planes, policies, values = _chunk_on_device(path, device)
            with torch.no_grad():
                for i in range(0, len(values) - batch_size + 1, batch_size):
                    p, v = net.forward(planes[i:i + batch_size])
                    policy_loss = calculate_policy_loss(F.softmax(p, dim=1), policies[i:i + batch_size])
                    value_loss = calculate_value_loss(v, values[i:i + batch_size])
                    loss = (policy_loss + SL_VALUE_WEIGHT * value_loss).mean()
                    sums += torch.stack([loss, policy_loss.mean(), value_loss.mean()]).double()
                    test_iteration += 1
        total = sums.tolist()
        test_loss = {"loss": total[0], "policy": total[1], "value": total[2]}
        n = max(test_iteration, 1)
        print(f"Test {epoch} : loss = {total[0] / n:6f}, time = {time.time() - started:3f} seconds.",
              file=sys.stderr)
        print(f"\tpolicy loss : {total[1] / n:6f}", file=sys.stderr)
        print(f"\tvalue loss  : {total[2] / n:6f}", file=sys.stderr)

        if epoch in LEARNING_SCHEDULE["learning_rate"]:
            previous_lr, current_lr = current_lr, LEARNING_SCHEDULE["learning_rate"][epoch]
            for group in optimizer.param_groups:
                group["lr"] = current_lr
            graphed = None
            print(f"Epoch {epoch}, learning rate has changed {previous_lr} -> {current_lr}")

    os.makedirs("model", exist_ok=True)
    torch.save(net.state_dict(), os.path.join("model", "sl-model.bin"))
    return test_loss
