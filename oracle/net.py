"""Oracle DualNet forward (TEST INFRASTRUCTURE - see oracle/__init__.py).

Restates nn/network/dual_net.py:17-106, res_block.py:8-38, head/policy_head.py:7-39
and head/value_head.py:7-39 as a functional PyTorch-CPU fp32 forward over a plain
``state_dict`` with the reference's key names (nn/utility.py:139-159 loads exactly
these).  Inference mode only: BatchNorm uses running statistics.
"""
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

FILTERS = 64      # dual_net.py:26
BLOCKS = 6        # dual_net.py:27
EPS_STEM = 1e-5   # nn.BatchNorm2d default, dual_net.py:32
EPS_BODY = 2e-5   # res_block.py:22-23, policy_head.py:19, value_head.py:20


def state_dict_shapes(board_size: int) -> Dict[str, Tuple[int, ...]]:
    """Key -> shape of every tensor ``load_network`` expects (nn/utility.py:139-159)."""
    p = board_size * board_size
    shapes = {"conv_layer.weight": (FILTERS, 6, 3, 3)}

    def bn(prefix, c):
        shapes[prefix + ".weight"] = (c,)
        shapes[prefix + ".bias"] = (c,)
        shapes[prefix + ".running_mean"] = (c,)
        shapes[prefix + ".running_var"] = (c,)
        shapes[prefix + ".num_batches_tracked"] = ()

    bn("bn_layer", FILTERS)
    for b in range(BLOCKS):
        shapes[f"blocks.{b}.conv1.weight"] = (FILTERS, FILTERS, 3, 3)
        shapes[f"blocks.{b}.conv2.weight"] = (FILTERS, FILTERS, 3, 3)
        bn(f"blocks.{b}.bn1", FILTERS)
        bn(f"blocks.{b}.bn2", FILTERS)
    shapes["policy_head.conv_layer.weight"] = (2, FILTERS, 1, 1)
    bn("policy_head.bn_layer", 2)
    shapes["policy_head.fc_layer.weight"] = (p + 1, 2 * p)
    shapes["policy_head.fc_layer.bias"] = (p + 1,)
    shapes["value_head.conv_layer.weight"] = (1, FILTERS, 1, 1)
    bn("value_head.bn_layer", 1)
    shapes["value_head.fc_layer.weight"] = (3, p)
    shapes["value_head.fc_layer.bias"] = (3,)
    return shapes


def make_state_dict(board_size: int, seed: int, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights (no trained weights exist - model/.gitkeep only).
    Only ``RandomState.random_sample`` plus exact float64 arithmetic is used so the
    values are bit-identical on every machine.  BN statistics are non-trivial on
    purpose (mean != 0, var != 1) so that folding errors show up."""
    rs = np.random.RandomState(seed)
    out = {}
    for key, shape in state_dict_shapes(board_size).items():
        if key.endswith("num_batches_tracked"):
            out[key] = torch.tensor(0, dtype=torch.long)
            continue
        n = int(np.prod(shape)) if shape else 1
        u = rs.random_sample(n)
        if key.endswith("running_var"):
            v = 0.5 + u                                  # [0.5, 1.5)
        elif key.endswith("running_mean"):
            v = (u - 0.5) * 0.2
        elif ".bn" in key or key.startswith("bn_layer"):
            v = 0.75 + 0.5 * u if key.endswith("weight") else (u - 0.5) * 0.2
        elif key.endswith("fc_layer.bias"):
            v = (u - 0.5) * 0.2
        else:
            fan_in = int(np.prod(shape[1:]))
            v = (u - 0.5) * 2.0 * gain * (3.0 / fan_in) ** 0.5
        out[key] = torch.from_numpy(v.astype(np.float32).reshape(shape))
    return out


def _bn(x, sd, prefix, eps):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, eps)


def forward_logits(sd: Dict[str, torch.Tensor], planes: torch.Tensor):
    """DualNet.forward, dual_net.py:41-52."""
    x = F.relu(_bn(F.conv2d(planes, sd["conv_layer.weight"], padding=1), sd, "bn_layer",
                   EPS_STEM))
    for b in range(BLOCKS):                                          # res_block.py:36-39
        h = F.relu(_bn(F.conv2d(x, sd[f"blocks.{b}.conv1.weight"], padding=1), sd,
                       f"blocks.{b}.bn1", EPS_BODY))
        h = _bn(F.conv2d(h, sd[f"blocks.{b}.conv2.weight"], padding=1), sd,
                f"blocks.{b}.bn2", EPS_BODY)
        x = F.relu(x + h)
    batch = x.shape[0]
    ph = F.relu(_bn(F.conv2d(x, sd["policy_head.conv_layer.weight"]), sd,
                    "policy_head.bn_layer", EPS_BODY))               # policy_head.py:33-39
    policy = F.linear(ph.reshape(batch, -1), sd["policy_head.fc_layer.weight"],
                      sd["policy_head.fc_layer.bias"])
    vh = F.relu(_bn(F.conv2d(x, sd["value_head.conv_layer.weight"]), sd,
                    "value_head.bn_layer", EPS_BODY))                # value_head.py:33-39
    value = F.linear(vh.reshape(batch, -1), sd["value_head.fc_layer.weight"],
                     sd["value_head.fc_layer.bias"])
    return policy, value


class OracleNet:
    """The two inference entry points the search uses (dual_net.py:81-106), CPU only."""

    def __init__(self, state_dict: Dict[str, torch.Tensor]):
        self.sd = state_dict

    def inference(self, planes: torch.Tensor):
        with torch.no_grad():
            policy, value = forward_logits(self.sd, planes)
            return torch.softmax(policy, dim=1), torch.softmax(value, dim=1)

    def inference_with_policy_logits(self, planes: torch.Tensor):
        with torch.no_grad():
            policy, value = forward_logits(self.sd, planes)
            return policy, torch.softmax(value, dim=1)
