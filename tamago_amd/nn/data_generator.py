"""Training-data generation from game records (mirror of nn/data_generator.py:17-149):
``sl_data_<k>.npz`` (every position x 8 symmetries, one-hot move targets) and
``rl_data_<k>.npz`` (8 random positions per self-play game, one random symmetry each,
improved-policy targets from the SGF comments) with the reference's keys, dtypes, chunking
and random-number call order (``random.shuffle`` of the file list, two
``np.random.permutation`` calls per game) - the files ``train.py`` consumes.

MI355X form: games are replayed on the host board only to collect position descriptors
(cells, side to move, previous move, move count, symmetry); the input planes of a whole
chunk are then produced by ONE launch of the featurise kernel (tg_featurize_sym_dev)
instead of one Python ``generate_input_planes`` per sample."""
import glob
import os
import random
from typing import List

import numpy as np

from tamago_amd.board.go_board import GoBoard
from tamago_amd.board.stone import Stone, color_value
from tamago_amd.nn.feature import featurize_batch, generate_rl_target_data, generate_target_data
from tamago_amd.sgf.reader import SGFReader

BATCH_SIZE = 256                       # learning_param.py:11
DATA_SET_SIZE = BATCH_SIZE * 4000      # learning_param.py:31


class _Samples:
    """Position descriptors + targets waiting to be written."""

    def __init__(self, size: int):
        self.size = size
        self.cells: List[np.ndarray] = []
        self.to_move: List[int] = []
        self.prev_move: List[int] = []
        self.moves: List[int] = []
        self.sym: List[int] = []
        self.policy: List[np.ndarray] = []
        self.value: List[int] = []

    def __len__(self):
        return len(self.value)

    def add(self, board: GoBoard, color, sym: int, policy: np.ndarray, value: int):
        self.cells.append(np.array(board.get_board_data(), dtype=np.uint8))
        self.to_move.append(color_value(color))
        self.prev_move.append(board.prev_move(1))
        self.moves.append(board.moves)
        self.sym.append(int(sym))
        self.policy.append(policy)
        self.value.append(value)

    def take(self, count: int) -> "_Samples":
        head = _Samples(self.size)
        for name in ("cells", "to_move", "prev_move", "moves", "sym", "policy", "value"):
            values = getattr(self, name)
            setattr(head, name, values[:count])
            setattr(self, name, values[count:])
        return head

    def planes(self, device_index: int = 0) -> np.ndarray:
        out = featurize_batch(self.size, np.stack(self.cells), np.array(self.to_move), np.array(self.prev_move),
                              np.array(self.moves), np.array(self.sym), device_index)
        return out.cpu().numpy()


def _save_data(save_file_path: str, samples: _Samples, kifu_counter: int) -> None:
    """nn/data_generator.py:17-34 (np.savez_compressed; value int32, kifu_count 0-d)."""
    np.savez_compressed(save_file_path, input=samples.planes(), policy=np.array(samples.policy),
                        value=np.array(samples.value, dtype=np.int32), kifu_count=np.array(kifu_counter))


def _write_chunks(program_dir: str, prefix: str, games, board_size: int, per_game) -> None:
    """Chunking of nn/data_generator.py:70-86 / :133-149: a full DATA_SET_SIZE chunk is written
    as soon as enough samples exist, the tail in whole mini-batches."""
    pending = _Samples(board_size)
    kifu_counter, data_counter = 1, 0
    for path in games:
        per_game(path, pending)
        if len(pending) >= DATA_SET_SIZE:
            _save_data(os.path.join(program_dir, "data", f"{prefix}_{data_counter}"),
                       pending.take(DATA_SET_SIZE), kifu_counter)
            kifu_counter = 1
            data_counter += 1
        kifu_counter += 1
    n_batches = len(pending) // BATCH_SIZE
    if n_batches > 0:
        _save_data(os.path.join(program_dir, "data", f"{prefix}_{data_counter}"),
                   pending.take(n_batches * BATCH_SIZE), kifu_counter)


def generate_supervised_learning_data(program_dir: str, kifu_dir: str, board_size: int = 9) -> None:
    """nn/data_generator.py:37-86."""
    board = GoBoard(board_size=board_size)

    def per_game(path: str, pending: _Samples):
        board.clear()
        sgf = SGFReader(path, board_size)
        color = Stone.BLACK
        value_label = sgf.get_value_label()
        for pos in sgf.get_moves():
            for sym in range(8):
                pending.add(board, color, sym, generate_target_data(board, pos, sym), value_label)
            board.put_stone(pos, color)
            color = Stone.get_opponent_color(color)
            value_label = 2 - value_label                # label is from the mover's point of view

    _write_chunks(program_dir, "sl_data", sorted(glob.glob(os.path.join(kifu_dir, "*.sgf"))), board_size, per_game)


def generate_reinforcement_learning_data(program_dir: str, kifu_dir_list: List[str], board_size: int = 9) -> None:
    """nn/data_generator.py:89-149."""
    board = GoBoard(board_size=board_size)
    kifu_list = []
    for kifu_dir in kifu_dir_list:
        kifu_list.extend(glob.glob(os.path.join(kifu_dir, "*.sgf")))
    random.shuffle(kifu_list)

    def per_game(path: str, pending: _Samples):
        board.clear()
        sgf = SGFReader(path, board_size)
        color = Stone.BLACK
        value_label = sgf.get_value_label()
        targets = set(int(i) for i in np.random.permutation(np.arange(sgf.get_n_moves()))[:8])
        sym_order = np.random.permutation(np.arange(8))
        taken = 0
        for i, pos in enumerate(sgf.get_moves()):
            if i in targets:
                sym = int(sym_order[taken])
                pending.add(board, color, sym, generate_rl_target_data(board, sgf.get_comment(i), sym), value_label)
                taken += 1
            board.put_stone(pos, color)
            color = Stone.get_opponent_color(color)
            value_label = 2 - value_label

    _write_chunks(program_dir, "rl_data", kifu_list, board_size, per_game)
