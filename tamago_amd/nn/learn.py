"""Training step for the DualNet the search evaluates (nn/learn.py:318-403, 126-232,
nn/loss.py:33-55).

The loop, the losses, the optimiser and the file formats interchange with the reference
(``model/rl-model.bin`` is the same ``state_dict``, ``model/rl-state.ckpt`` holds a
``torch.optim.SGD`` state over the parameters in the same order).  The mini-batch step is ``HipTrainer`` = this repo's HIP
kernels (tamago_amd/csrc/train.hip behind ``tg_trainer_*``: forward with batch statistics, backward, SGD-Nesterov, running
statistics) - the ONLY step implementation in the package: a board size the kernels are not built for (anything but 9x9
and 19x19) is refused with an error, not handed to a library.  The torch-autograd restatement the kernels are tested against
lives with the other checkers in ``oracle/train_ref.py``.  The arithmetic is fp32 throughout (the reference runs this step
under fp16 autocast with a GradScaler on the GPU, learn.py:342,371, and in fp32 on the CPU - fp32 is the stricter of the
two; mixed precision stays optional and unbuilt).

The network is held as a flat table of tensors keyed like the state_dict (no module tree):
the same table feeds ``DualNet.load_state_dict`` of the inference side after a step.
"""
import glob
import os
import sys
import time
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from tamago_amd.nn.network.dual_net import BLOCKS, random_state_dict, state_dict_keys

SL_LEARNING_RATE = 0.01      # learning_param.py
RL_LEARNING_RATE = 0.01
MOMENTUM = 0.9
WEIGHT_DECAY = 1e-4
SL_VALUE_WEIGHT = 0.02
RL_VALUE_WEIGHT = 1.0
EPOCHS = 15
LEARNING_SCHEDULE = {"learning_rate": {5: 0.001, 8: 0.0001, 10: 0.00001}}

_STEM_BN = (1e-5, 0.1)       # nn.BatchNorm2d defaults (dual_net.py:32)
_BODY_BN = (2e-5, 0.01)      # res_block.py:21-22, head/*.py:21


# ---------------------------------------------------------------------------------- losses
def calculate_policy_kld_loss(output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """loss.py:33-43: KL(target || softmax(output)), summed, divided by the batch size."""
    logp = F.log_softmax(output, dim=-1)
    return F.kl_div(logp, target, reduction="batchmean")


def calculate_policy_loss(output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """loss.py:9-19: cross entropy of a probability output against a distribution."""
    return -(target * torch.log(output.float() + 1e-8)).sum(dim=1)


def calculate_sl_policy_loss(output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """loss.py:21-31: per-sample cross entropy against a move class."""
    return F.cross_entropy(output, target, reduction="none")


def calculate_value_loss(output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """loss.py:45-55: per-sample cross entropy over (loss, draw, win)."""
    return F.cross_entropy(output, target, reduction="none")


# -------------------------------------------------------------------------------- the table
class ParamTable:
    """Parameters and batch-norm statistics of one DualNet as a flat table of tensors keyed like the reference's
    state_dict (no module tree, no forward pass).  ``parameters()`` yields the trainable ones in the reference module's
    order: a ``torch.optim.SGD`` built over them has the state layout of the reference's ``model/rl-state.ckpt``."""

    def __init__(self, device: torch.device, board_size: int = 9,
                 state: Dict[str, torch.Tensor] = None):
        self.device = torch.device(device)
        self.board_size = board_size
        self.training = True
        self.t: Dict[str, torch.Tensor] = {}
        self.load_state_dict(state if state is not None else random_state_dict(board_size))

    def load_state_dict(self, state: Dict[str, torch.Tensor]):
        table = {}
        for key, shape in state_dict_keys(self.board_size):
            v = torch.as_tensor(state[key]).detach().to(self.device, torch.float32).clone()
            if tuple(v.shape) != tuple(shape):
                raise ValueError(f"shape mismatch for {key}: {tuple(v.shape)} vs {shape}")
            if not key.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
            table[key] = v
        self.t = table
        tracked = state.get("bn_layer.num_batches_tracked")
        self.batches_tracked = int(tracked) if tracked is not None else 0

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """CPU copy with the reference's keys, plus the ``num_batches_tracked`` counters a
        torch module writes (one per batch norm, all equal to the number of training
        forwards; no computation reads them - every batch norm has a fixed momentum)."""
        out = {}
        for key, _ in state_dict_keys(self.board_size):
            out[key] = self.t[key].detach().to("cpu").clone()
            if key.endswith("running_var"):
                out[key[:-len("running_var")] + "num_batches_tracked"] = \
                    torch.tensor(self.batches_tracked)
        return out

    def parameters(self):
        return [v for v in self.t.values() if v.requires_grad]


def make_optimizer(net: ParamTable, lr: float) -> torch.optim.SGD:
    """learn.py:333-337: SGD, Nesterov momentum 0.9, weight decay 1e-4 on every parameter.  (The product path uses it for
    the optimiser-state FILE only - the update itself is train.hip's sgd kernels.)"""
    return torch.optim.SGD(net.parameters(), lr=lr, momentum=MOMENTUM,
                           weight_decay=WEIGHT_DECAY, nesterov=True)


class HipTrainer:
    """The training step as this repo's HIP kernels (tamago_amd/csrc/train.hip behind tg_trainer_*): forward with
    batch statistics, backward, SGD-Nesterov + weight decay, batch-norm running statistics - no autograd, no
    library kernel.  Same table / state_dict / optimiser-state layout as ParamTable + make_optimizer (and as the
    reference's modules).  fp32; built for 9x9 (the size every tuning decision was made at) and 19x19 (the same kernels walking
    the board in more passes) - any other board size raises (there is no second backend to fall to)."""

    def __init__(self, device: torch.device, board_size: int = 9, batch_size: int = 256,
                 state: Dict[str, torch.Tensor] = None):
        import ctypes
        from tamago_amd import lib as _lib
        self._lib_mod = _lib
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", 0)           # (tg_trainer_create below uses index 0 as well)
        if board_size not in (9, 19):
            raise ValueError(f"HipTrainer: the HIP training step (tamago_amd/csrc/train.hip) is built for 9x9 and 19x19 boards; "
                             f"board size {board_size} is not served (the reference's own trainer, nn/learn.py, is the "
                             f"tool for it)")
        self.board_size = board_size
        self.batch_size = batch_size
        self.keys = state_dict_keys(board_size)
        state = state if state is not None else random_state_dict(board_size)
        flat = np.concatenate([torch.as_tensor(state[k]).detach().to("cpu", torch.float32).numpy().reshape(-1)
                               for k, _ in self.keys]).astype(np.float32)
        self.n_params = flat.size
        tracked = state.get("bn_layer.num_batches_tracked")
        self.batches_tracked = int(tracked) if tracked is not None else 0
        handle = ctypes.c_void_p()
        index = self.device.index if self.device.index is not None else 0
        _lib.check(self.lib.tg_trainer_create(board_size, index, batch_size, flat.ctypes.data, flat.size,
                                              ctypes.byref(handle)), "tg_trainer_create")
        self.handle = handle
        self.steps = 0

    def close(self):
        if getattr(self, "handle", None) is not None:
            self.lib.tg_trainer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, plane: torch.Tensor, policy: torch.Tensor, value: torch.Tensor, mode: str = "rl",
             lr: float = RL_LEARNING_RATE):
        """Enqueue one mini-batch on the current stream (device tensors, batch = batch_size)."""
        assert plane.is_cuda and plane.shape == (self.batch_size, 6, self.board_size, self.board_size)
        if plane.device != self.device or policy.device != self.device or value.device != self.device:
            raise ValueError(f"HipTrainer on {self.device}: step() got tensors on {plane.device} / {policy.device} / {value.device}")
        plane = plane.contiguous().float()
        policy = policy.contiguous().float()
        value = value.contiguous().long()
        weight = RL_VALUE_WEIGHT if mode == "rl" else SL_VALUE_WEIGHT
        # the trainer's own device is made current for the launch (a caller on cuda:0 driving a trainer on cuda:1 would
        # otherwise enqueue onto the wrong device's stream)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self._lib_mod.check(self.lib.tg_trainer_step(self.handle, plane.data_ptr(), policy.data_ptr(), value.data_ptr(),
                                                         int(mode != "rl"), float(weight), float(lr), stream),
                                "tg_trainer_step")
        self._keep = (plane, policy, value)
        self.steps += 1
        self.batches_tracked += 1

    def take_losses(self) -> Dict[str, float]:
        """Summed losses since the last call (one host read)."""
        sums = np.zeros(3, dtype=np.float64)
        self._lib_mod.check(self.lib.tg_trainer_read_losses(self.handle, sums.ctypes.data, 1), "tg_trainer_read_losses")
        return {"loss": float(sums[0]), "policy": float(sums[1]), "value": float(sums[2])}

    def _blobs(self):
        params = np.zeros(self.n_params, dtype=np.float32)
        mom = np.zeros(self.n_params, dtype=np.float32)
        self._lib_mod.check(self.lib.tg_trainer_get_params(self.handle, params.ctypes.data, mom.ctypes.data,
                                                           self.n_params), "tg_trainer_get_params")
        return params, mom

    def _unflatten(self, flat):
        out, o = {}, 0
        for key, shape in self.keys:
            n = int(np.prod(shape))
            out[key] = torch.from_numpy(flat[o:o + n].reshape(shape).copy())
            o += n
        return out

    def state_dict(self) -> Dict[str, torch.Tensor]:
        table = self._unflatten(self._blobs()[0])
        out = {}
        for key, _ in self.keys:
            out[key] = table[key]
            if key.endswith("running_var"):
                out[key[:-len("running_var")] + "num_batches_tracked"] = torch.tensor(self.batches_tracked)
        return out

    def momentum_buffers(self):
        """Momentum buffers of the trainable tensors in parameters() order (torch.optim.SGD state layout)."""
        table = self._unflatten(self._blobs()[1])
        return [table[k] for k, _ in self.keys if not k.endswith(("running_mean", "running_var"))]

    def load_momentum_buffers(self, buffers):
        flat, it = [], iter(buffers)
        for key, shape in self.keys:
            if key.endswith(("running_mean", "running_var")):
                flat.append(np.zeros(int(np.prod(shape)), dtype=np.float32))
            else:
                flat.append(torch.as_tensor(next(it)).detach().to("cpu", torch.float32).numpy().reshape(-1))
        blob = np.concatenate(flat).astype(np.float32)
        self._lib_mod.check(self.lib.tg_trainer_set_momentum(self.handle, blob.ctypes.data, blob.size),
                            "tg_trainer_set_momentum")


# ------------------------------------------------------------------------------ file formats
def load_data_set(path: str):
    """nn/utility.py:90-102 - one shuffle of the chunk from numpy's global generator."""
    data = np.load(path)
    perm = np.random.permutation(len(data["value"]))
    return (data["input"][perm], data["policy"][perm].astype(np.float32),
            data["value"][perm].astype(np.int64))


def print_learning_process(loss_data, epoch, index, iteration, start_time):
    """nn/utility.py:43-59 (stderr, same three lines)."""
    n = max(iteration, 1)
    spent = time.time() - start_time
    print(f"epoch {epoch}, data-{index} : loss = {loss_data['loss'] / n:6f}, "
          f"time = {spent:3f} seconds.", file=sys.stderr)
    print(f"\tpolicy loss : {loss_data['policy'] / n:6f}", file=sys.stderr)
    print(f"\tvalue loss  : {loss_data['value'] / n:6f}", file=sys.stderr)


def train_with_gumbel_alphazero_on_gpu(program_dir: str, board_size: int, batch_size: int,
                                       device_index: int = 0) -> Dict[str, float]:
    """learn.py:318-403: one pass over ``data/rl_data_*.npz``, resuming from and writing
    ``model/rl-model.bin`` / ``model/rl-state.ckpt``.  Returns the summed losses of the
    last chunk (the reference returns nothing)."""
    if not torch.cuda.is_available():
        raise RuntimeError("tamago_amd trains on the GPU only")
    device = torch.device("cuda", device_index)
    torch.cuda.set_device(device)             # graph capture and side streams run on the CURRENT device
    data_set = sorted(glob.glob(os.path.join(program_dir, "data", "rl_data_*.npz")))
    net = ParamTable(device, board_size)
    model_file_path = os.path.join(program_dir, "model", "rl-model.bin")
    if os.path.exists(model_file_path):
        print(f"load {model_file_path}")
        net.load_state_dict(torch.load(model_file_path, map_location="cpu"))
    optimizer = make_optimizer(net, RL_LEARNING_RATE)
    num_trained_batches = 0
    # fp32 needs no loss scaling; the entry exists because the reference's GPU trainer reads it
    # on resume (learn.py:356) - a fresh GradScaler's state, or the one a loaded checkpoint had
    scaler_state = {"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5,
                    "growth_interval": 2000, "_growth_tracker": 0}
    state_file_path = os.path.join(program_dir, "model", "rl-state.ckpt")
    if os.path.exists(state_file_path):
        print(f"load {state_file_path}")
        checkpoint = torch.load(state_file_path, map_location=device)
        optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
        num_trained_batches = checkpoint["num_trained_batches"]
        scaler_state = checkpoint.get("scaler_state_dict", scaler_state)
        print(f"num_trained_batches : {num_trained_batches}")

    train_loss = {"loss": 0.0, "policy": 0.0, "value": 0.0}
    hip = HipTrainer(device, board_size, batch_size, net.state_dict())      # (refuses sizes train.hip is not built for)
    buffers = [optimizer.state[p].get("momentum_buffer") for p in net.parameters()]
    if all(b is not None for b in buffers):
        hip.load_momentum_buffers(buffers)
    for data_index, path in enumerate(data_set):
        plane_data, policy_data, value_data = load_data_set(path)
        planes = torch.from_numpy(plane_data).to(device, torch.float32)   # chunk resident in HBM
        policies = torch.from_numpy(policy_data).to(device)
        values = torch.from_numpy(value_data).to(device)
        train_loss = {"loss": 0.0, "policy": 0.0, "value": 0.0}
        iteration = 0
        started = time.time()
        for i in range(0, len(value_data) - batch_size + 1, batch_size):
            hip.step(planes[i:i + batch_size], policies[i:i + batch_size], values[i:i + batch_size],
                     mode="rl", lr=RL_LEARNING_RATE)
            num_trained_batches += 1
            iteration += 1
        train_loss = hip.take_losses()
        print_learning_process(train_loss, 0, data_index, iteration, started)

    # back into the table / torch.optim.SGD layout the files are written from
    net.load_state_dict(hip.state_dict())
    fresh = make_optimizer(net, RL_LEARNING_RATE)            # (load_state_dict made new parameter tensors)
    had_momentum = any("momentum_buffer" in st for st in optimizer.state.values())
    if hip.steps or had_momentum:
        for p, buf in zip(net.parameters(), hip.momentum_buffers()):
            fresh.state[p]["momentum_buffer"] = buf.to(device)
    optimizer = fresh
    hip.close()
    os.makedirs(os.path.dirname(model_file_path), exist_ok=True)
    torch.save(net.state_dict(), model_file_path)
    torch.save({"num_trained_batches": num_trained_batches,
                "optimizer_state_dict": optimizer.state_dict(),
                "scaler_state_dict": scaler_state}, state_file_path)
    return train_loss


def split_train_test_set(file_list, train_data_ratio: float):
    """nn/utility.py:105-122: the first 80 % of the chunk files train, the rest test."""
    cut = int(len(file_list) * train_data_ratio)
    train_files, test_files = file_list[:cut], file_list[cut:]
    print(f"Training data set : {train_files}")
    print(f"Testing data set  : {test_files}")
    return train_files, test_files


def _chunk_on_device(path: str, device):
    plane_data, policy_data, value_data = load_data_set(path)
    return (torch.from_numpy(plane_data).to(device, torch.float32),
            torch.from_numpy(policy_data).to(device), torch.from_numpy(value_data).to(device))


def train_on_gpu(program_dir: str, board_size: int, batch_size: int, epochs: int,
                 device_index: int = 0) -> Dict[str, float]:
    """Supervised trainer, learn.py:126-232: `epochs` passes over the training chunks of
    ``data/sl_data_*.npz``, after each one the test chunks in eval mode, then the learning
    rate schedule; writes ``model/sl-model.bin`` relative to the working directory, as the
    reference does (learn.py:232).  Returns the last test-loss sums.

    Training steps: ``HipTrainer`` (mode "sl": cross entropy of the softmax output against the move distribution, value
    weight 0.02; the learning rate is an argument of every step, so the schedule needs no re-capture).  Test pass: the
    trained table goes into the INFERENCE network (``DualNet``: batch norm folded from the running statistics = eval
    mode) and the losses are taken from its policy / value probabilities."""
    if not torch.cuda.is_available():
        raise RuntimeError("tamago_amd trains on the GPU only")
    from tamago_amd.nn.network.dual_net import DualNet
    device = torch.device("cuda", device_index)
    torch.cuda.set_device(device)
    data_set = sorted(glob.glob(os.path.join(program_dir, "data", "sl_data_*.npz")))
    train_files, test_files = split_train_test_set(data_set, 0.8)
    hip = HipTrainer(device, board_size, batch_size, random_state_dict(board_size))
    evaluator = DualNet(device, board_size)
    current_lr = SL_LEARNING_RATE
    test_loss = {"loss": 0.0, "policy": 0.0, "value": 0.0}
    for epoch in range(epochs):
        for data_index, path in enumerate(train_files):
            planes, policies, values = _chunk_on_device(path, device)
            iteration = 0
            started = time.time()
            for i in range(0, len(values) - batch_size + 1, batch_size):
                hip.step(planes[i:i + batch_size], policies[i:i + batch_size], values[i:i + batch_size],
                         mode="sl", lr=current_lr)
                iteration += 1
            print_learning_process(hip.take_losses(), epoch, data_index, iteration, started)

        sums = torch.zeros(3, dtype=torch.float64, device=device)       # accumulated on the device, read once per epoch
        test_iteration = 0
        started = time.time()
        evaluator.load_state_dict(hip.state_dict())
        for path in test_files:
            planes, policies, values = _chunk_on_device(path, device)
            for i in range(0, len(values) - batch_size + 1, batch_size):
                prob, vprob = evaluator.forward_device(planes[i:i + batch_size].contiguous())
                policy_loss = calculate_policy_loss(prob, policies[i:i + batch_size])
                target = values[i:i + batch_size].long()
                # = cross entropy of the value logits; a probability that underflowed to 0 must not turn the epoch's sum into inf
                value_loss = -torch.log(vprob.gather(1, target[:, None])[:, 0].clamp_min(1e-38))
                loss = (policy_loss + SL_VALUE_WEIGHT * value_loss).mean()
                sums += torch.stack([loss, policy_loss.mean(), value_loss.mean()]).double()
                test_iteration += 1
        total = sums.cpu().tolist()
        test_loss = {"loss": total[0], "policy": total[1], "value": total[2]}
        n = max(test_iteration, 1)
        print(f"Test {epoch} : loss = {total[0] / n:6f}, time = {time.time() - started:3f} seconds.",
              file=sys.stderr)
        print(f"\tpolicy loss : {total[1] / n:6f}", file=sys.stderr)
        print(f"\tvalue loss  : {total[2] / n:6f}", file=sys.stderr)

        if epoch in LEARNING_SCHEDULE["learning_rate"]:
            previous_lr, current_lr = current_lr, LEARNING_SCHEDULE["learning_rate"][epoch]
            print(f"Epoch {epoch}, learning rate has changed {previous_lr} -> {current_lr}")

    os.makedirs("model", exist_ok=True)
    torch.save(hip.state_dict(), os.path.join("model", "sl-model.bin"))
    hip.close()
    return test_loss
