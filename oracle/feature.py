"""Oracle feature planes (TEST INFRASTRUCTURE - see oracle/__init__.py).

Restates nn/feature.py:10-57 (generate_input_planes, sym = 0 as used by search).
"""
import numpy as np

from oracle.board import GoBoard, PASS, WHITE


def symmetric_index(size: int, q: int, sym: int) -> int:
    """On-board index (row-major) that output point q reads under symmetry `sym`
    (board/go_board.py:80-104 sym_map, :499-509 get_symmetrical_coordinate)."""
    y, x = divmod(q, size)
    n = size - 1
    sy, sx = [(y, x), (y, n - x), (n - y, x), (n - y, n - x),
              (x, y), (n - x, y), (x, n - y), (n - x, n - y)][sym]
    return sy * size + sx


def generate_input_planes(board: GoBoard, color: int, sym: int = 0) -> np.ndarray:
    """float32[6,S,S], row-major from the top-left on-board point:
    0 empty, 1 own stones, 2 opponent stones (colours swapped for WHITE to move,
    feature.py:24-25), 3 one-hot previous move (:43-45), 4 all-ones iff
    ``moves > 1`` and the previous move was PASS (:39-41, plane 3 then zero),
    5 side to move (+1 black / -1 white, :50-52)."""
    size = board.get_board_size()
    n = size * size
    perm = [symmetric_index(size, q, sym) for q in range(n)]
    cells = np.array(board.get_board_data(), dtype=np.int64)[perm]
    if color == WHITE:
        cells = np.where(cells == 0, 0, 3 - cells)
    planes = np.zeros((6, n), dtype=np.float32)
    planes[0] = cells == 0
    planes[1] = cells == 1
    planes[2] = cells == 2
    previous = board.record_pos(board.moves - 1)          # record.py:65-74
    if board.moves > 1 and previous == PASS:
        planes[4] = 1.0
    else:
        planes[3] = np.array([1.0 if previous == board.onboard_pos[perm[q]] else 0.0
                              for q in range(n)], dtype=np.float32)
    planes[5] = -1.0 if color == WHITE else 1.0
    return planes.reshape(6, size, size)
