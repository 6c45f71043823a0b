"""Coordinate conversion with the conventions of board/coordinate.py:6-82: internal positions
index the padded board (one border cell each side), GTP columns skip the letter I, SGF uses
a..s from the top-left corner, PASS / RESIGN have their own spellings."""
from tamago_amd.board.constant import PASS, RESIGN, OB_SIZE, GTP_X_COORDINATE

_SGF_LETTERS = "abcdefghijklmnopqrstuvwxyz"
_SPECIAL_TO_GTP = {PASS: "pass", RESIGN: "resign"}
_SPECIAL_FROM_GTP = {"PASS": PASS, "RESIGN": RESIGN}


class Coordinate:
    def __init__(self, board_size: int):
        self.board_size = board_size
        self.board_size_with_ob = board_size + OB_SIZE * 2

    def _column_row(self, pos: int):
        """(column, row) of an internal position, both 0-based on the playable board, row 0 on top."""
        row, column = divmod(pos, self.board_size_with_ob)
        return column - OB_SIZE, row - OB_SIZE

    def convert_from_gtp_format(self, pos: str) -> int:
        text = pos.upper()
        if text in _SPECIAL_FROM_GTP:
            return _SPECIAL_FROM_GTP[text]
        column = GTP_X_COORDINATE.index(text[0]) - 1
        row = self.board_size - int(text[1:])
        return (row + OB_SIZE) * self.board_size_with_ob + column + OB_SIZE

    def convert_to_gtp_format(self, pos: int) -> str:
        if pos in _SPECIAL_TO_GTP:
            return _SPECIAL_TO_GTP[pos]
        column, row = self._column_row(pos)
        return f"{GTP_X_COORDINATE[column + 1]}{self.board_size - row}"

    def convert_to_sgf_format(self, pos: int) -> str:
        if pos in _SPECIAL_TO_GTP:
            return "tt"
        column, row = self._column_row(pos)
        return _SGF_LETTERS[column] + _SGF_LETTERS[row]
