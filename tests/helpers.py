"""Shared helpers for the parity tests (oracle side)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def oracle_replay(size, moves, colors, upto, superko=False):
    from oracle.board import GoBoard
    board = GoBoard(size, 7.0, superko)
    for mv, c in zip(moves[:upto], colors[:upto]):
        board.put_stone(int(mv), int(c))
    return board


def unhex(lst):
    return np.array([float.fromhex(v) for v in lst], dtype=np.float64)
