"""RL / SL data generator (next row 8(f).3) vs npz files written by the reference's
nn/data_generator.py on the reference-recorded self-play games (tools/gen_golden_datagen.py);
input planes come from the HIP featurise kernel, so this needs a GPU."""
import glob
import hashlib
import os
import random

import numpy as np
import pytest

from tests.helpers import load_json, load_npz

pytestmark = pytest.mark.gpu


def _write_games(root, one_dir_per_game):
    games = load_json("selfplay_games.json")
    dirs = []
    for key in sorted(games):
        d = os.path.join(root, "g" + key.replace(",", "_")) if one_dir_per_game else os.path.join(root, "all")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, key.split(",")[0] + ".sgf"), "w", encoding="utf-8") as f:
            f.write(games[key])
        dirs.append(d)
    return dirs


def test_rl_data_files_equal_reference(tmp_path, monkeypatch):
    import tamago_amd.nn.data_generator as dg
    meta = load_json("datagen_s9.json")
    fix = load_npz("datagen_s9.npz")
    monkeypatch.setattr(dg, "BATCH_SIZE", meta["rl_batch_size"])
    monkeypatch.setattr(dg, "DATA_SET_SIZE", meta["rl_data_set_size"])
    dirs = _write_games(str(tmp_path), True)
    os.makedirs(tmp_path / "prog" / "data")
    random.seed(meta["rl_seed"])
    np.random.seed(meta["rl_seed"])
    dg.generate_reinforcement_learning_data(str(tmp_path / "prog"), dirs, 9)
    files = sorted(os.path.basename(f) for f in glob.glob(str(tmp_path / "prog" / "data" / "rl_data_*.npz")))
    assert files == meta["rl_files"]
    for name in files:
        got = np.load(tmp_path / "prog" / "data" / name)
        stem = name[:-4]
        for key in ("input", "policy", "value", "kifu_count"):
            want = fix[f"{stem}_{key}"]
            assert got[key].dtype == want.dtype and got[key].shape == want.shape, (name, key)
            assert np.array_equal(got[key], want), (name, key)


def test_sl_data_files_equal_reference(tmp_path, monkeypatch):
    import tamago_amd.nn.data_generator as dg
    meta = load_json("datagen_s9.json")
    fix = load_npz("datagen_s9.npz")
    monkeypatch.setattr(dg, "BATCH_SIZE", meta["sl_batch_size"])
    monkeypatch.setattr(dg, "DATA_SET_SIZE", meta["sl_data_set_size"])
    dirs = _write_games(str(tmp_path), False)
    os.makedirs(tmp_path / "prog" / "data")
    dg.generate_supervised_learning_data(str(tmp_path / "prog"), dirs[0], 9)
    files = sorted(os.path.basename(f) for f in glob.glob(str(tmp_path / "prog" / "data" / "sl_data_*.npz")))
    assert files == sorted(meta["sl_files"])
    for name in files:
        got = np.load(tmp_path / "prog" / "data" / name)
        for key, want in meta["sl_files"][name].items():
            a = np.ascontiguousarray(got[key])
            assert list(got[key].shape) == want["shape"] and str(a.dtype) == want["dtype"], (name, key)
            assert hashlib.sha256(a.tobytes()).hexdigest() == want["sha256"], (name, key)
        stem = name[:-4]
        assert np.array_equal(got["input"][:16], fix[f"{stem}_input_head"])
        assert np.array_equal(got["policy"][:16], fix[f"{stem}_policy_head"])
