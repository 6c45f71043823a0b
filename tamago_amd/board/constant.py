"""Board constants (mirror of board/constant.py:4-31; the board size is a runtime
parameter here instead of a module constant)."""
OB_SIZE = 1
PASS = 0
RESIGN = -1
GTP_X_COORDINATE = 'IABCDEFGHJKLMNOPQRSTUVWXYZ'


def max_records(board_size: int) -> int:
    """board/constant.py:31."""
    return board_size * board_size * 3
