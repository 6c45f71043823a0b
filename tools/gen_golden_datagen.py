#!/usr/bin/env python3
"""Golden fixtures for the "next" row 8(f).3 (SGF reader + RL / SL data generator), produced by
running the REFERENCE's sgf/reader.py and nn/data_generator.py on the reference-recorded
self-play games of tests/golden/selfplay_games.json.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tools/gen_golden_datagen.py

The reference's chunk sizes (learning_param.py: BATCH_SIZE 256, DATA_SET_SIZE 1 024 000) are
patched down IN THE IMPORTED MODULE so that three games exercise the chunking and the
remainder path; the patched values are recorded in the fixture.  RL games sit in one directory
each (kifu_dir_list order is then the list order - a plain glob order is file-system
dependent and feeds random.shuffle)."""
import glob
import hashlib
import json
import os
import random
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")


def digest(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def main():
    import nn.data_generator as dg                      # the reference
    from sgf.reader import SGFReader
    games = json.load(open(os.path.join(GOLD, "selfplay_games.json")))
    keys = sorted(games)
    work = tempfile.mkdtemp(prefix="dg_")
    dirs = []
    for k in keys:
        d = os.path.join(work, "g" + k.replace(",", "_"))
        os.makedirs(d)
        with open(os.path.join(d, k.split(",")[0] + ".sgf"), "w", encoding="utf-8") as f:
            f.write(games[k])
        dirs.append(d)
    alld = os.path.join(work, "all")
    os.makedirs(alld)
    for k in keys:
        with open(os.path.join(alld, k.split(",")[0] + ".sgf"), "w", encoding="utf-8") as f:
            f.write(games[k])

    # --- reader ---
    reader = {}
    for k in keys:
        r = SGFReader(games[k], 9, literal=True)
        reader[k] = {
            "size": r.size, "komi": r.komi, "value_label": r.get_value_label(), "n_moves": r.get_n_moves(),
            "moves": [int(r.get_move_data(i)) for i in range(r.get_n_moves())],
            "colors": [int(r.get_color(i).value) for i in range(r.get_n_moves())],
            "comment_sha": [hashlib.sha256(r.get_comment(i).encode()).hexdigest()[:16] for i in range(r.get_n_moves())],
            "comment0": r.get_comment(0),
            "application": r.application, "black": r.black_player_name, "white": r.white_player_name,
        }
    # reader edge cases: passes written as "B[]" and "B[tt]", ignored tags, handicap-less header
    text = "(;GM[1]FF[4]SZ[9]KM[6.5]RE[W+R]PB[a]PW[b]DT[2024-01-01];B[ee]C[first];W[];B[tt]C[x y:1];W[aa])"
    r = SGFReader(text, 9, literal=True)
    reader["edge"] = {"text": text, "komi": r.komi, "value_label": r.get_value_label(),
                      "moves": [int(r.get_move_data(i)) for i in range(r.get_n_moves())],
                      "comments": [r.get_comment(i) for i in range(r.get_n_moves())]}

    # --- RL data ---
    dg.BATCH_SIZE, dg.DATA_SET_SIZE = 8, 16
    prog = os.path.join(work, "prog")
    os.makedirs(os.path.join(prog, "data"))
    random.seed(11)
    np.random.seed(11)
    dg.generate_reinforcement_learning_data(prog, dirs, 9)
    out = {}
    meta = {"rl_batch_size": 8, "rl_data_set_size": 16, "rl_seed": 11, "keys": keys, "reader": reader}
    files = sorted(glob.glob(os.path.join(prog, "data", "rl_data_*.npz")))
    meta["rl_files"] = [os.path.basename(f) for f in files]
    for f in files:
        z = np.load(f)
        name = os.path.basename(f)[:-4]
        for key in ("input", "policy", "value", "kifu_count"):
            out[f"{name}_{key}"] = z[key]
    # --- SL data (digests: 8 symmetries of every position) ---
    dg.BATCH_SIZE, dg.DATA_SET_SIZE = 64, 1024
    prog2 = os.path.join(work, "prog2")
    os.makedirs(os.path.join(prog2, "data"))
    dg.generate_supervised_learning_data(prog2, alld, 9)
    files = sorted(glob.glob(os.path.join(prog2, "data", "sl_data_*.npz")))
    meta["sl_batch_size"], meta["sl_data_set_size"] = 64, 1024
    meta["sl_files"] = {}
    for f in files:
        z = np.load(f)
        meta["sl_files"][os.path.basename(f)] = {
            key: {"shape": list(z[key].shape), "dtype": str(z[key].dtype), "sha256": digest(z[key])}
            for key in ("input", "policy", "value", "kifu_count")}
        name = os.path.basename(f)[:-4]
        out[f"{name}_input_head"] = z["input"][:16]
        out[f"{name}_policy_head"] = z["policy"][:16]
        out[f"{name}_value_head"] = z["value"][:16]
    np.savez_compressed(os.path.join(GOLD, "datagen_s9.npz"), **out)
    json.dump(meta, open(os.path.join(GOLD, "datagen_s9.json"), "w"), indent=1)
    print("wrote datagen_s9.npz / .json:", meta["rl_files"], list(meta["sl_files"]))


if __name__ == "__main__":
    sys.exit(main())
