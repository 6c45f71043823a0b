"""Host-side mirror of the two nn/utility.py functions on the hot path."""
import torch

from tamago_amd.nn.network.dual_net import DualNet


def get_torch_device(use_gpu: bool, device_index: int = 0) -> torch.device:
    """nn/utility.py:12-24.  The reference hard-wires cuda:0; a self-play shard passes its
    own index (one process per GPU).  The product path has no CPU mode."""
    if not use_gpu:
        raise RuntimeError("tamago_amd is the GPU path; use the reference for CPU inference")
    torch.cuda.set_device(device_index)
    return torch.device("cuda", device_index)


def load_network(model_file_path: str, use_gpu: bool, board_size: int = 9,
                 device_index: int = 0) -> DualNet:
    """nn/utility.py:139-159: build, try to load the state_dict, keep the random
    initialisation (with the same message) if that fails, inference mode."""
    device = get_torch_device(use_gpu, device_index)
    network = DualNet(device, board_size)
    try:
        network.load_state_dict(torch.load(model_file_path, map_location="cpu"))
    except Exception:  # the reference uses a bare except
        print(f"Failed to load {model_file_path}.")
    network.eval()
    torch.set_grad_enabled(False)
    return network
