"""Coordinate conversion (mirror of board/coordinate.py:6-82)."""
from tamago_amd.board.constant import PASS, RESIGN, OB_SIZE, GTP_X_COORDINATE


class Coordinate:
    def __init__(self, board_size: int):
        self.board_size = board_size
        self.board_size_with_ob = board_size + OB_SIZE * 2

    def convert_from_gtp_format(self, pos: str) -> int:
        if pos.upper() == "PASS":
            return PASS
        if pos.upper() == "RESIGN":
            return RESIGN
        x = GTP_X_COORDINATE.index(pos.upper()[0]) - 1
        y = self.board_size - int(pos[1:])
        return x + OB_SIZE + (y + OB_SIZE) * self.board_size_with_ob

    def convert_to_gtp_format(self, pos: int) -> str:
        if pos == PASS:
            return "pass"
        if pos == RESIGN:
            return "resign"
        x = pos % self.board_size_with_ob - OB_SIZE + 1
        y = self.board_size - (pos // self.board_size_with_ob - OB_SIZE)
        return GTP_X_COORDINATE[x] + str(y)

    def convert_to_sgf_format(self, pos: int) -> str:
        if pos in (PASS, RESIGN):
            return "tt"
        letters = "abcdefghijklmnopqrstuvwxyz"
        x = pos % self.board_size_with_ob - OB_SIZE
        y = pos // self.board_size_with_ob - OB_SIZE
        return letters[x] + letters[y]
