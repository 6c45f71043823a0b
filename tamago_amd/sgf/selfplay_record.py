"""Self-play game record (output format of sgf/selfplay_record.py:45-110): moves plus the
improved-policy comment ``"<n> <gtp>:<p:.3e> ..."`` that the reference's RL data generator
reads back, so a shard's ``.sgf`` files can be consumed by the reference's ``train.py --rl``."""
import os

from tamago_amd.board.coordinate import Coordinate
from tamago_amd.board.stone import Stone, color_value

PROGRAM_NAME = "TamaGo"     # program.py:3 - the data generator does not care, kept for identical files


class SelfPlayRecord:
    def __init__(self, save_dir: str, coord: Coordinate):
        self.save_dir = save_dir
        self.coord = coord
        self.file_index = 1
        self.clear()

    def clear(self):
        self.colors = []
        self.moves = []
        self.comments = []

    def set_index(self, index: int):
        self.file_index = index

    def save_record(self, root, pos: int, color):
        """selfplay_record.py:45-65."""
        improved = root.calculate_improved_policy()
        parts = [str(root.get_num_children())]
        for i in range(root.get_num_children()):
            parts.append(f"{self.coord.convert_to_gtp_format(root.get_child_move(i))}:{improved[i]:.3e}")
        self.colors.append(color_value(color))
        self.moves.append(self.coord.convert_to_sgf_format(pos))
        self.comments.append(" ".join(parts))

    def to_sgf(self, winner, komi: float, is_resign: bool, score: float) -> str:
        """selfplay_record.py:67-104."""
        text = f"(;FF[4]GM[1]SZ[{self.coord.board_size}]\n"
        text += f"AP[{PROGRAM_NAME}]PB[{PROGRAM_NAME}-Black]PW[{PROGRAM_NAME}-White]"
        if winner is Stone.BLACK:
            text += "RE[B+R]" if is_resign else f"RE[B+{score:.1f}]"
        elif winner is Stone.WHITE:
            text += "RE[W+R]" if is_resign else f"RE[W+{-score:.1f}]"
        else:
            text += "RE[0]"
        text += f"KM[{komi}]"
        for color, move, comment in zip(self.colors, self.moves, self.comments):
            text += f";{'B' if color == 1 else 'W'}[{move}]C[{comment}]"
        return text + "\n)"

    def write_record(self, winner, komi: float, is_resign: bool, score: float):
        path = os.path.join(self.save_dir, f"{self.file_index}.sgf")
        with open(path, mode="w", encoding="utf-8") as out:
            out.write(self.to_sgf(winner, komi, is_resign, score))
        self.file_index += 1
