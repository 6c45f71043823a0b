"""Feature / target construction (mirror of nn/feature.py:10-102).

``generate_input_planes`` runs the HIP featurise kernel (tg_featurize_sym_dev) - for bulk
work call the kernel through ``featurize_batch``; the two target builders are host integer /
string logic exactly as in the reference."""
import numpy as np
import torch

from tamago_amd import lib as _lib
from tamago_amd.board.constant import PASS
from tamago_amd.board.go_board import GoBoard
from tamago_amd.board.stone import color_value


def symmetric_index(size: int, q: int, sym: int) -> int:
    """On-board index that output point q reads under symmetry `sym` (go_board.py:80-104)."""
    y, x = divmod(q, size)
    n = size - 1
    sy, sx = [(y, x), (y, n - x), (n - y, x), (n - y, n - x),
              (x, y), (n - x, y), (x, n - y), (n - x, n - y)][sym]
    return sy * size + sx


def featurize_batch(size: int, cells: np.ndarray, to_move: np.ndarray, prev_move: np.ndarray,
                    moves: np.ndarray, sym: np.ndarray = None, device_index: int = 0) -> torch.Tensor:
    """cells uint8 [B,P], to_move [B], prev_move [B], moves [B], sym [B] -> CUDA fp32 [B,6,S,S]."""
    lib = _lib.load()
    dev = torch.device("cuda", device_index)
    b = cells.shape[0]
    d_cells = torch.as_tensor(np.ascontiguousarray(cells, dtype=np.uint8)).to(dev)
    d_tm = torch.as_tensor(np.ascontiguousarray(to_move, dtype=np.int8)).to(dev)
    d_prev = torch.as_tensor(np.ascontiguousarray(prev_move, dtype=np.int32)).to(dev)
    d_moves = torch.as_tensor(np.ascontiguousarray(moves, dtype=np.int32)).to(dev)
    d_sym = torch.as_tensor(np.ascontiguousarray(sym, dtype=np.int8)).to(dev) if sym is not None else None
    out = torch.empty((b, 6, size, size), dtype=torch.float32, device=dev)
    _lib.check(lib.tg_featurize_sym_dev(size, d_cells.data_ptr(), d_tm.data_ptr(), d_prev.data_ptr(),
                                        d_moves.data_ptr(), d_sym.data_ptr() if d_sym is not None else None,
                                        b, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
               "tg_featurize_sym_dev")
    return out


def generate_input_planes(board: GoBoard, color, sym: int = 0) -> np.ndarray:
    """nn/feature.py:10-57 for one position (float32 [6,S,S] on the host)."""
    size = board.get_board_size()
    out = featurize_batch(size, np.array([board.get_board_data()], dtype=np.uint8),
                          np.array([color_value(color)]), np.array([board.prev_move(1)]),
                          np.array([board.moves]), np.array([sym]))
    return out[0].cpu().numpy()


def generate_target_data(board: GoBoard, target_pos: int, sym: int = 0) -> np.ndarray:
    """nn/feature.py:60-77: one-hot move label over the symmetric board + PASS slot."""
    size = board.get_board_size()
    onboard = board.onboard_pos
    target = [1 if target_pos == onboard[symmetric_index(size, q, sym)] else 0 for q in range(size * size)]
    target.append(1 if target_pos == PASS else 0)
    return np.array(target)


def generate_rl_target_data(board: GoBoard, improved_policy_data: str, sym: int = 0) -> np.ndarray:
    """nn/feature.py:80-102: improved-policy comment "<n> <gtp>:<p> ..." -> target vector
    (1e-18 for moves that were not candidates), symmetric board order + PASS slot."""
    size = board.get_board_size()
    onboard = board.onboard_pos
    table = [1e-18] * len(board.cells)
    for item in improved_policy_data.split(" ")[1:]:
        name, prob = item.split(":")
        table[board.coordinate.convert_from_gtp_format(name)] = float(prob)
    target = [table[onboard[symmetric_index(size, q, sym)]] for q in range(size * size)]
    target.append(table[PASS])
    return np.array(target)
