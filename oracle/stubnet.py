"""Deterministic stand-in evaluator (TEST INFRASTRUCTURE - see oracle/__init__.py).

Tree-parity tests must hold the network outputs bit-identical between the reference,
the oracle and the HIP tree engine on ANY machine (a 1e-7 change
in the policy already changes visit counts).  A real convolution cannot promise that,
so these tests use this evaluator instead: a pure integer hash of the input planes
turned into float32 by single IEEE operations (int -> float32 conversion and one
division), which every platform rounds identically.  It mimics the DualNet inference
API of nn/network/dual_net.py:81-106; the outputs are deliberately peaky so that the
searches go deep.
"""
import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def plane_hash(planes: np.ndarray) -> np.ndarray:
    """uint64 hash per position of float32 planes [B,6,S,S] (entries are -1, 0, 1)."""
    b = planes.shape[0]
    q = np.rint(planes.reshape(b, -1)).astype(np.int64) + 2           # 1, 2, 3
    n = q.shape[1]
    with np.errstate(over="ignore"):
        w = _mix(np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        h = (q.astype(np.uint64) * w[None, :]).sum(axis=1, dtype=np.uint64)
    return _mix(h)


def stub_outputs(planes: np.ndarray, salt: int = 0):
    """Returns (policy float32[B,A] summing to ~1, logits float32[B,A],
    value float32[B,3] summing to ~1)."""
    b = planes.shape[0]
    a = planes.shape[2] * planes.shape[3] + 1
    h = plane_hash(planes)
    with np.errstate(over="ignore"):
        ha = _mix(h[:, None] + np.arange(1, a + 1, dtype=np.uint64)[None, :]
                  * np.uint64(0xD1B54A32D192ED03) + np.uint64(salt))
    base = (ha % np.uint64(64)).astype(np.int64) + 1
    spike = np.where((ha >> np.uint64(8)) % np.uint64(11) == 0, 3000, 0).astype(np.int64)
    k = base + spike                                                  # 1 .. 3064
    total = k.sum(axis=1, keepdims=True)
    policy = k.astype(np.float32) / total.astype(np.float32)
    logits = ((ha >> np.uint64(20)) % np.uint64(1024)).astype(np.float32) / np.float32(128.0) \
        - np.float32(4.0)
    hv = _mix(h[:, None] + np.arange(101, 104, dtype=np.uint64)[None, :] + np.uint64(salt))
    kv = (hv % np.uint64(1000)).astype(np.int64) + 1
    value = kv.astype(np.float32) / kv.sum(axis=1, keepdims=True).astype(np.float32)
    return policy, logits, value


class StubNet:
    """Drop-in for DualNet.inference / inference_with_policy_logits."""

    def __init__(self, salt: int = 0):
        self.salt = salt
        self.calls = []

    def inference(self, planes: torch.Tensor):
        policy, _, value = stub_outputs(planes.numpy(), self.salt)
        self.calls.append(planes.shape[0])
        return torch.from_numpy(policy), torch.from_numpy(value)

    def inference_with_policy_logits(self, planes: torch.Tensor):
        _, logits, value = stub_outputs(planes.numpy(), self.salt)
        self.calls.append(planes.shape[0])
        return torch.from_numpy(logits), torch.from_numpy(value)
