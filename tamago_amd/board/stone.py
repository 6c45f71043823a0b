"""Cell colours with the values of board/stone.py:5-27 (the device code uses the same numbers:
0 empty, 1 black, 2 white, 3 border)."""
from enum import Enum


class Stone(Enum):
    EMPTY, BLACK, WHITE, OUT_OF_BOARD = range(4)

    @classmethod
    def get_opponent_color(cls, color):
        """BLACK <-> WHITE; anything else comes back unchanged."""
        return _OTHER_SIDE.get(color, color)


_OTHER_SIDE = {Stone.BLACK: Stone.WHITE, Stone.WHITE: Stone.BLACK}


def color_value(color) -> int:
    """Accepts a Stone or a plain int (1 black / 2 white)."""
    return color.value if isinstance(color, Stone) else int(color)
