"""Oracle feature planes (TEST INFRASTRUCTURE - see oracle/__init__.py).

Restates nn/feature.py:10-57 (generate_input_planes, sym = 0 as used by search).
"""
import numpy as np

from oracle.board import GoBoard, PASS, WHITE


def generate_input_planes(board: GoBoard, color: int) -> np.ndarray:
    """float32[6,S,S], row-major from the top-left on-board point:
    0 empty, 1 own stones, 2 opponent stones (colours swapped for WHITE to move,
    feature.py:24-25), 3 one-hot previous move (:43-45), 4 all-ones iff
    ``moves > 1`` and the previous move was PASS (:39-41, plane 3 then zero),
    5 side to move (+1 black / -1 white, :50-52)."""
    size = board.get_board_size()
    n = size * size
    cells = np.array(board.get_board_data(), dtype=np.int64)
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
        planes[3] = np.array([1.0 if previous == p else 0.0 for p in board.onboard_pos],
                             dtype=np.float32)
    planes[5] = -1.0 if color == WHITE else 1.0
    return planes.reshape(6, size, size)
