"""DualNet on MI355X - host-side mirror of the reference class (nn/network/dual_net.py:13-106).

Same constructor, ``load_state_dict`` keys (nn/utility.py:139-159) and the two inference
entry points the search uses; the arithmetic runs in the fused HIP kernel behind
``tg_net_forward_*`` (tamago_amd/csrc/net_forward.hip).  There is no PyTorch fallback.
"""
import ctypes
from typing import Dict, Tuple

import numpy as np
import torch

from tamago_amd import lib as _lib

FILTERS = 64
BLOCKS = 6


def state_dict_keys(board_size: int):
    """(key, shape) in the order tg_net_create expects (include/tamago_hip.h)."""
    p = board_size * board_size
    keys = [("conv_layer.weight", (FILTERS, 6, 3, 3))]

    def bn(prefix, c):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            keys.append((f"{prefix}.{leaf}", (c,)))

    bn("bn_layer", FILTERS)
    for b in range(BLOCKS):
        keys.append((f"blocks.{b}.conv1.weight", (FILTERS, FILTERS, 3, 3)))
        keys.append((f"blocks.{b}.conv2.weight", (FILTERS, FILTERS, 3, 3)))
        bn(f"blocks.{b}.bn1", FILTERS)
        bn(f"blocks.{b}.bn2", FILTERS)
    keys.append(("policy_head.conv_layer.weight", (2, FILTERS, 1, 1)))
    bn("policy_head.bn_layer", 2)
    keys.append(("policy_head.fc_layer.weight", (p + 1, 2 * p)))
    keys.append(("policy_head.fc_layer.bias", (p + 1,)))
    keys.append(("value_head.conv_layer.weight", (1, FILTERS, 1, 1)))
    bn("value_head.bn_layer", 1)
    keys.append(("value_head.fc_layer.weight", (3, p)))
    keys.append(("value_head.fc_layer.bias", (3,)))
    return keys


def random_state_dict(board_size: int) -> Dict[str, torch.Tensor]:
    """Random initialisation with the distributions torch.nn uses by default for the
    reference's layers (Conv2d / Linear: U(+-1/sqrt(fan_in)); BatchNorm: identity), drawn
    from torch's global generator.  This is what ``load_network`` keeps when the model
    file cannot be read (nn/utility.py:152-155)."""
    sd = {}
    for key, shape in state_dict_keys(board_size):
        if key.endswith("running_mean"):
            sd[key] = torch.zeros(shape)
        elif key.endswith("running_var"):
            sd[key] = torch.ones(shape)
        elif "bn" in key.split(".")[-2]:
            sd[key] = torch.ones(shape) if key.endswith("weight") else torch.zeros(shape)
        else:
            w_shape = shape if key.endswith("weight") else None
            if w_shape is None:      # fc bias: fan_in of the matching weight
                fan_in = 2 * board_size ** 2 if key.startswith("policy") else board_size ** 2
            else:
                fan_in = int(np.prod(shape[1:]))
            bound = 1.0 / fan_in ** 0.5
            sd[key] = (torch.rand(shape) * 2.0 - 1.0) * bound
    return sd


class DualNet:
    """``DualNet(device, board_size)`` as in dual_net.py:16-17; ``device`` is a
    torch.device (``cuda`` / ``cuda:N``)."""

    def __init__(self, device: torch.device, board_size: int = 9):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.TamagoHipError("tamago_amd.DualNet runs on the GPU only (device='cuda')")
        self.board_size = board_size
        self.device_index = self.device.index if self.device.index is not None else 0
        self._lib = _lib.load()
        self._handle = None
        self._state = random_state_dict(board_size)
        self._upload()

    # -- torch.nn.Module look-alikes used by nn/utility.py:139-159 ---------------------------
    def to(self, device):
        return self

    def eval(self):
        return self

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return dict(self._state)

    def load_state_dict(self, state: Dict[str, torch.Tensor]):
        new = {}
        for key, shape in state_dict_keys(self.board_size):
            if key not in state:
                raise KeyError(f"missing key in state_dict: {key}")
            t = torch.as_tensor(state[key]).detach().to("cpu", torch.float32)
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"shape mismatch for {key}: {tuple(t.shape)} vs {shape}")
            new[key] = t.contiguous()
        self._state = new
        self._upload()

    def _upload(self):
        flat = np.concatenate([self._state[k].numpy().reshape(-1)
                               for k, _ in state_dict_keys(self.board_size)]).astype(np.float32)
        assert flat.size == self._lib.tg_net_param_count(self.board_size)
        import ctypes
        handle = ctypes.c_void_p()
        _lib.check(self._lib.tg_net_create(self.board_size, self.device_index,
                                           flat.ctypes.data, flat.size, ctypes.byref(handle)),
                   "tg_net_create")
        if self._handle is not None:
            self._lib.tg_net_destroy(self._handle)
        self._handle = handle

    def __del__(self):
        try:
            if self._handle is not None:
                self._lib.tg_net_destroy(self._handle)
        except Exception:
            pass

    @property
    def handle(self):
        return self._handle

    # -- inference -----------------------------------------------------------------------
    def range_fallbacks(self) -> int:
        """Forward launches of this network that the exact-fp32 kernel had to redo because a layer output left the f16 range
        of the split-operand kernel (tg_net_range_fallbacks; synchronises the device).  0 for a healthy network."""
        count = ctypes.c_ulonglong(0)
        _lib.check(self._lib.tg_net_range_fallbacks(self._handle, ctypes.byref(count)), "tg_net_range_fallbacks")
        return int(count.value)

    def range_fallback_positions(self) -> int:
        """Positions the exact-fp32 kernel redid in those launches (tg_net_range_fallback_positions): the one-axis Winograd kernels
        mark the workgroup passes that left the range, only those are redone."""
        count = ctypes.c_ulonglong(0)
        _lib.check(self._lib.tg_net_range_fallback_positions(self._handle, ctypes.byref(count)), "tg_net_range_fallback_positions")
        return int(count.value)

    def band_timeouts(self) -> int:
        """19x19: bounded waits of the banded forward kernels that gave up (tg_net_band_timeouts; synchronises the device).
        Each one is also a range fallback; after the first the network keeps to the one-workgroup kernel."""
        count = ctypes.c_ulonglong(0)
        _lib.check(self._lib.tg_net_band_timeouts(self._handle, ctypes.byref(count)), "tg_net_band_timeouts")
        return int(count.value)

    def set_shared_device(self, shared: bool = True) -> None:
        """Other processes drive this GPU too (more self-play shards than GPUs): kernels that need several workgroups of one
        launch resident at once are not chosen (tg_net_set_shared_device).  Results do not change."""
        _lib.check(self._lib.tg_net_set_shared_device(self._handle, 1 if shared else 0), "tg_net_set_shared_device")

    def _forward_host(self, input_plane: torch.Tensor, want_logits: int):
        x = input_plane.detach().to("cpu", torch.float32).contiguous()
        b, s = x.shape[0], self.board_size
        if tuple(x.shape[1:]) != (6, s, s):
            raise ValueError(f"expected [B,6,{s},{s}], got {tuple(x.shape)}")
        policy = torch.empty((b, s * s + 1), dtype=torch.float32)
        value = torch.empty((b, 3), dtype=torch.float32)
        _lib.check(self._lib.tg_net_forward_host(self._handle, x.data_ptr(), b, want_logits,
                                                 policy.data_ptr(), value.data_ptr()),
                   "tg_net_forward_host")
        return policy, value

    def inference(self, input_plane: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """softmax(policy), softmax(value) on the host - dual_net.py:81-91."""
        return self._forward_host(input_plane, 0)

    def inference_with_policy_logits(self, input_plane: torch.Tensor):
        """raw policy logits, softmax(value) - dual_net.py:94-106."""
        return self._forward_host(input_plane, 1)

    def forward_device(self, planes: torch.Tensor, want_logits: bool = False, out=None):
        """Device-resident variant: planes is a CUDA fp32 tensor [B,6,S,S]; enqueues on the
        current torch stream and returns CUDA tensors (no host hop)."""
        assert planes.is_cuda and planes.dtype == torch.float32 and planes.is_contiguous()
        b, s = planes.shape[0], self.board_size
        if tuple(planes.shape[1:]) != (6, s, s):
            # the reference's conv / fc layers raise a shape error here; the device kernels would
            # index a [B, S*S+1] policy with another board's stride
            raise ValueError(f"expected [B,6,{s},{s}], got {tuple(planes.shape)}")
        if out is None:
            policy = torch.empty((b, self.board_size ** 2 + 1), dtype=torch.float32,
                                 device=planes.device)
            value = torch.empty((b, 3), dtype=torch.float32, device=planes.device)
        else:
            policy, value = out
        stream = torch.cuda.current_stream(planes.device).cuda_stream
        _lib.check(self._lib.tg_net_forward_dev(self._handle, planes.data_ptr(), b,
                                                int(want_logits), policy.data_ptr(),
                                                value.data_ptr(), stream),
                   "tg_net_forward_dev")
        return policy, value
