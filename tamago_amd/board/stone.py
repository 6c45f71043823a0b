"""Cell colours (mirror of board/stone.py:5-27)."""
from enum import Enum


class Stone(Enum):
    EMPTY = 0
    BLACK = 1
    WHITE = 2
    OUT_OF_BOARD = 3

    @classmethod
    def get_opponent_color(cls, color):
        if color == Stone.BLACK:
            return Stone.WHITE
        if color == Stone.WHITE:
            return Stone.BLACK
        return color


def color_value(color) -> int:
    """Accepts a Stone or a plain int (1 black / 2 white)."""
    return color.value if isinstance(color, Stone) else int(color)
