"""The closed loop of SURVEY §8(f) rows 1, 3 and 4 on the device: self-play records -> RL data
chunks -> training step -> the trained weights drive the next generation's search."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_two_generations(tmp_path, monkeypatch):
    import rl_loop
    import tamago_amd.nn.data_generator as dg
    monkeypatch.setattr(dg, "BATCH_SIZE", 32)
    torch.manual_seed(11)
    np.random.seed(11)
    prog = str(tmp_path)
    lines = []
    s0, l0 = rl_loop.run_generation(prog, 0, 24, 16, 16, 32, log=lines.append)
    w0 = torch.load(os.path.join(prog, "model", "rl-model.bin"), map_location="cpu")
    s1, l1 = rl_loop.run_generation(prog, 1, 24, 16, 16, 32, log=lines.append)
    w1 = torch.load(os.path.join(prog, "model", "rl-model.bin"), map_location="cpu")
    assert s0["games"] == s1["games"] == 24
    assert len(glob.glob(os.path.join(prog, "archive", "0", "*.sgf"))) == 24
    assert len(glob.glob(os.path.join(prog, "archive", "1", "*.sgf"))) == 24
    chunk = np.load(glob.glob(os.path.join(prog, "data", "rl_data_*.npz"))[0])
    assert chunk["input"].shape[1:] == (6, 9, 9) and len(chunk["value"]) % 32 == 0
    assert np.isfinite(l0["loss"]) and np.isfinite(l1["loss"])
    ck = torch.load(os.path.join(prog, "model", "rl-state.ckpt"), map_location="cpu")
    assert ck["num_trained_batches"] == 2 * (24 * 8 // 32)     # 8 sampled positions per game
    moved = max(float((w1[k].float() - w0[k].float()).abs().max()) for k in w0 if w0[k].dtype.is_floating_point)
    assert moved > 1e-4
