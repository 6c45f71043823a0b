"""Host-side Go board for the game loop (one move per search, not on the hot path).

Keeps the reference GoBoard's public surface used by the search callers
(board/go_board.py: put_stone, is_legal, get_all_legal_pos, get_board_data,
count_score, copy_board, record access).  Internally it is a plain flood-fill board -
per-descent board work happens on the GPU (tamago_amd/csrc/search.hip), which receives
this object's state through ``tg_search_set_root``.
"""
from typing import List

import numpy as np

from tamago_amd.board.constant import PASS, max_records
from tamago_amd.board.coordinate import Coordinate
from tamago_amd.board.stone import Stone, color_value

EMPTY, BLACK, WHITE, OOB = 0, 1, 2, 3

_ZOBRIST = {}


def zobrist_keys(board_size: int) -> np.ndarray:
    """uint64 [4][(S+2)^2] (board/zobrist_hash.py:9-10, but from a private generator so
    the global numpy stream the search relies on is not touched)."""
    if board_size not in _ZOBRIST:
        rs = np.random.RandomState(0x5A0B + board_size)
        n = (board_size + 2) ** 2
        hi = rs.randint(0, 2 ** 32, size=(4, n)).astype(np.uint64)
        lo = rs.randint(0, 2 ** 32, size=(4, n)).astype(np.uint64)
        _ZOBRIST[board_size] = np.ascontiguousarray((hi << np.uint64(32)) | lo)
    return _ZOBRIST[board_size]


class GoBoard:
    def __init__(self, board_size: int, komi: float = 7.0, check_superko: bool = False):
        self.board_size = board_size
        self.board_size_with_ob = board_size + 2
        self.komi = komi
        self.check_superko = check_superko
        w = self.board_size_with_ob
        self.onboard_pos = [x + y * w for y in range(1, board_size + 1)
                            for x in range(1, board_size + 1)]
        self.coordinate = Coordinate(board_size)
        self.zobrist = zobrist_keys(board_size)
        self.max_records = max_records(board_size)
        self.clear()

    # ---------------------------------------------------------------------------------
    def clear(self):
        w = self.board_size_with_ob
        self.cells = np.full(w * w, OOB, dtype=np.uint8)
        self.cells[self.onboard_pos] = EMPTY
        self.moves = 1
        self.ko_move = 0
        self.ko_pos = 0
        self.prisoner = [0, 0]
        self.hash = 0
        self.rec_color = [0] * self.max_records
        self.rec_pos = [PASS] * self.max_records
        self.rec_hash = np.zeros(self.max_records, dtype=np.uint64)
        self.handicap_pos: List[int] = []

    def get_board_size(self) -> int:
        return self.board_size

    def get_komi(self) -> float:
        return self.komi

    def set_komi(self, komi: float):
        self.komi = komi

    def get_neighbor4(self, pos: int) -> List[int]:
        w = self.board_size_with_ob
        return [pos - w, pos - 1, pos + 1, pos + w]

    def _string(self, pos: int):
        """(stones, liberties) of the string at pos by flood fill."""
        color = self.cells[pos]
        stones, libs, stack = {pos}, set(), [pos]
        while stack:
            p = stack.pop()
            for n in self.get_neighbor4(p):
                c = self.cells[n]
                if c == EMPTY:
                    libs.add(n)
                elif c == color and n not in stones:
                    stones.add(n)
                    stack.append(n)
        return stones, libs

    def _save(self, color: int, pos: int):
        if self.moves < self.max_records:
            self.rec_color[self.moves] = color
            self.rec_pos[self.moves] = pos
            self.rec_hash[self.moves] = self.hash

    def put_stone(self, pos: int, color) -> None:
        """board/go_board.py:131-185 (no legality check, like the reference)."""
        color = color_value(color)
        if pos == PASS:
            self._save(color, pos)
            self.moves += 1
            return
        other = 3 - color
        self.cells[pos] = color
        self.hash ^= int(self.zobrist[color][pos])
        captured = 0
        for n in self.get_neighbor4(pos):
            if self.cells[n] == other:
                stones, libs = self._string(n)
                if not libs:
                    for p in stones:
                        self.cells[p] = EMPTY
                        self.hash ^= int(self.zobrist[other][p])
                    captured += len(stones)
        self.prisoner[color - 1] += captured
        if captured == 1 and all(self.cells[n] != color for n in self.get_neighbor4(pos)):
            libs = [n for n in self.get_neighbor4(pos) if self.cells[n] == EMPTY]
            if len(libs) == 1:
                self.ko_move = self.moves
                self.ko_pos = libs[0]
        self._save(color, pos)
        self.moves += 1

    def put_handicap_stone(self, pos: int, color) -> None:
        """board/go_board.py:187-235: a stone like any other (captures, prisoners, ko, positional hash) that is no MOVE -
        the move counter and the move record stay as they are, the point goes to the handicap list."""
        color = color_value(color)
        other = 3 - color
        self.cells[pos] = color
        self.hash ^= int(self.zobrist[color][pos])
        captured = 0
        for n in self.get_neighbor4(pos):
            if self.cells[n] == other:
                stones, libs = self._string(n)
                if not libs:
                    for p in stones:
                        self.cells[p] = EMPTY
                        self.hash ^= int(self.zobrist[other][p])
                    captured += len(stones)
        self.prisoner[color - 1] += captured
        if captured == 1 and all(self.cells[n] != color for n in self.get_neighbor4(pos)):
            libs = [n for n in self.get_neighbor4(pos) if self.cells[n] == EMPTY]
            if len(libs) == 1:
                self.ko_move = self.moves
                self.ko_pos = libs[0]
        self.handicap_pos.append(pos)

    def get_handicap_history(self) -> List[int]:
        """board/go_board.py:546-552."""
        return self.handicap_pos[:]

    def is_legal(self, pos: int, color) -> bool:
        """board/go_board.py:260-304."""
        color = color_value(color)
        other = 3 - color
        if self.cells[pos] != EMPTY:
            return False
        nbrs = self.get_neighbor4(pos)
        info = {}
        for n in nbrs:
            if self.cells[n] in (BLACK, WHITE):
                info[n] = self._string(n)
        if not any(self.cells[n] == EMPTY for n in nbrs):
            alive = False
            for n in nbrs:
                if self.cells[n] == other and len(info[n][1]) == 1:
                    alive = True
                if self.cells[n] == color and len(info[n][1]) > 1:
                    alive = True
            if not alive:
                return False
        if self.ko_pos == pos and self.ko_move == self.moves - 1:
            return False
        if self.check_superko:
            h = self.hash
            seen = []
            for n in nbrs:
                if n in info and len(info[n][1]) == 1 and not any(n in s for s in seen):
                    seen.append(info[n][0])
                    for p in info[n][0]:
                        h ^= int(self.zobrist[other][p])     # reference quirk: opponent keys
            h ^= int(self.zobrist[color][pos])
            if np.any(self.rec_hash == np.uint64(h)):
                return False
        return True

    def get_all_legal_pos(self, color) -> List[int]:
        return [p for p in self.onboard_pos if self.is_legal(p, color)]

    def get_board_data(self, sym: int = 0) -> List[int]:
        if sym != 0:
            raise NotImplementedError("symmetries are a training-time feature (out of scope)")
        return [int(self.cells[p]) for p in self.onboard_pos]

    def get_to_move(self) -> Stone:
        if self.moves == 1:
            return Stone.BLACK
        return Stone.get_opponent_color(Stone(self.rec_color[self.moves - 1]))

    def count_score(self) -> int:
        """board/go_board.py:561-608 with its quirks (atari stones dead; an empty point is
        coloured by its direct neighbours only; colourings feed later points)."""
        work = [int(v) for v in self.cells]
        for pos in self.onboard_pos:
            if self.cells[pos] in (BLACK, WHITE) and len(self._string(pos)[1]) == 1:
                work[pos] = EMPTY
        for pos in self.onboard_pos:
            if work[pos] != EMPTY:
                continue
            color = EMPTY
            for n in self.get_neighbor4(pos):
                v = work[n]
                if v in (BLACK, WHITE):
                    if color == EMPTY:
                        color = v
                    elif color != v:
                        color = OOB
            work[pos] = color
        return work.count(BLACK) - work.count(WHITE)

    # -- state handed to the GPU search ---------------------------------------------------
    def prev_move(self, back: int = 1) -> int:
        idx = self.moves - back
        return self.rec_pos[idx] if 0 <= idx < self.max_records else PASS


def copy_board(dst: GoBoard, src: GoBoard) -> None:
    """board/go_board.py:611-626."""
    dst.cells = src.cells.copy()
    dst.moves = src.moves
    dst.ko_move = src.ko_move
    dst.ko_pos = src.ko_pos
    dst.prisoner = src.prisoner[:]
    dst.hash = src.hash
    dst.rec_color = src.rec_color[:]
    dst.rec_pos = src.rec_pos[:]
    dst.rec_hash = src.rec_hash.copy()
    dst.handicap_pos = src.handicap_pos[:]
