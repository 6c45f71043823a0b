"""Handicap points (board/handicap.py:1-86): the reference keeps a table per board size; the points follow the usual rule and
are produced from it here - odd sizes 9 .. 19, the 3rd line up to 11x11 and the 4th beyond, 2 .. 9 stones, listed row by row
from the top, left to right (tests/golden/handicap.json holds the reference's table for comparison)."""
from typing import List, Optional

from tamago_amd.board.constant import GTP_X_COORDINATE

# which of the nine star points (row: 0 top, 1 middle, 2 bottom; column: 0 left, 1 centre, 2 right) n stones take
_PATTERN = {
    2: [(0, 2), (2, 0)],
    3: [(0, 0), (0, 2), (2, 0)],
    4: [(0, 0), (0, 2), (2, 0), (2, 2)],
    5: [(0, 0), (0, 2), (1, 1), (2, 0), (2, 2)],
    6: [(0, 0), (0, 2), (1, 0), (1, 2), (2, 0), (2, 2)],
    7: [(0, 0), (0, 2), (1, 0), (1, 1), (1, 2), (2, 0), (2, 2)],
    8: [(0, 0), (0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1), (2, 2)],
    9: [(r, c) for r in range(3) for c in range(3)],
}


def get_handicap_coordinates(size: int, handicaps: int) -> Optional[List[str]]:
    """GTP coordinates of `handicaps` fixed handicap stones on a size x size board, None when the reference has no entry."""
    if size % 2 == 0 or not 9 <= size <= 19 or handicaps not in _PATTERN:
        return None
    line = 3 if size <= 11 else 4
    cols = [line, (size + 1) // 2, size + 1 - line]             # 1-based from the left
    rows = [size + 1 - line, (size + 1) // 2, line]             # GTP row numbers: top, middle, bottom
    return [f"{GTP_X_COORDINATE[cols[c]]}{rows[r]}" for r, c in _PATTERN[handicaps]]
