"""Oracle Go board (TEST INFRASTRUCTURE - see oracle/__init__.py).

Restates the observable behaviour of the reference board engine
(/root/reference/board/go_board.py, string.py, pattern.py, record.py,
zobrist_hash.py) with different internals: strings are kept as Python sets of
stones / liberties instead of the reference's array-embedded sorted linked
lists.  Observable results (cell colours, captures, ko point, legality incl.
positional superko, self-atari size, eye detection, candidate order) are
identical; tests/test_oracle_board.py proves this against reference-generated
fixtures.

Coordinates follow the reference: 1-D padded board ``pos = x + y*(S+2)`` with a
one-cell OUT_OF_BOARD border, ``PASS = 0``, ``RESIGN = -1``
(board/constant.py:4-31, go_board.py:28-104).
"""
from typing import List

import numpy as np

EMPTY, BLACK, WHITE, OUT_OF_BOARD = 0, 1, 2, 3
PASS = 0
RESIGN = -1


def opponent(color: int) -> int:
    """board/stone.py:14-27 - BLACK<->WHITE, other values unchanged."""
    if color == BLACK:
        return WHITE
    if color == WHITE:
        return BLACK
    return color


# --------------------------------------------------------------------------------------
# 3x3 neighbourhood codes and the eye table (board/pattern.py)
# --------------------------------------------------------------------------------------
# pat3 code: 2 bits per neighbour, order NW,N,NE,W,E,SW,S,SE at bit offsets 0,2,...,14
# (pattern.py:10-19,47-50).  The reference keeps pat3 incrementally; since it is always
# equal to the true neighbour colours (border cells pre-marked 3, pattern.py:102-117) the
# oracle recomputes it from the cell array on demand.

def _build_eye_table() -> np.ndarray:
    """Eye table, pattern.py:52-98.

    The reference enumerates 22 base shapes x 8 symmetries x colour swap.  The
    resulting 90 BLACK + 90 WHITE codes are exactly characterised by this rule
    (verified entry-by-entry against the reference table in
    tests/test_oracle_board.py::test_eye_table):

    * the four orthogonal neighbours are own stones or border, the border forming
      nothing / one full side / one corner (the only geometrically possible cases);
    * centre (4 on-board diagonals): at most one opponent diagonal, or exactly two
      opponent diagonals with the other two being own stones;
    * side (2 on-board diagonals): at least one own diagonal, or both opponent;
    * corner (1 on-board diagonal): always.
    """
    table = np.zeros(65536, dtype=np.uint8)
    # neighbour slots: 0 NW 1 N 2 NE 3 W 4 E 5 SW 6 S 7 SE
    border_sets = [()]  # no border
    border_sets += [(0, 1, 2), (5, 6, 7), (0, 3, 5), (2, 4, 7)]          # N, S, W, E side
    border_sets += [(0, 1, 2, 3, 5), (0, 1, 2, 4, 7), (0, 3, 5, 6, 7), (2, 4, 5, 6, 7)]  # corners
    orth = (1, 3, 4, 6)
    diag = (0, 2, 5, 7)
    for border in border_sets:
        free_diag = [d for d in diag if d not in border]
        n_free = len(free_diag)
        for combo in range(3 ** n_free):
            cells = [0] * 8
            for b in border:
                cells[b] = OUT_OF_BOARD
            for o in orth:
                if o not in border:
                    cells[o] = BLACK
            c = combo
            n_own = n_opp = 0
            for d in free_diag:
                v = c % 3
                c //= 3
                cells[d] = v
                n_own += v == BLACK
                n_opp += v == WHITE
            if n_free == 4:
                ok = n_opp <= 1 or (n_opp == 2 and n_own == 2)
            elif n_free == 2:
                ok = n_own >= 1 or n_opp == 2
            else:
                ok = True
            if not ok:
                continue
            code = 0
            swapped = 0
            for k, v in enumerate(cells):
                code |= v << (2 * k)
                sv = v if v in (EMPTY, OUT_OF_BOARD) else 3 - v
                swapped |= sv << (2 * k)
            table[code] = BLACK
            table[swapped] = WHITE
    return table


EYE_TABLE = _build_eye_table()


def make_zobrist(board_size: int, seed: int = 0x7A6F62) -> np.ndarray:
    """Zobrist keys.  The reference draws them from the unseeded global numpy RNG at
    import time (zobrist_hash.py:9-10), so they differ per process and never influence
    results (up to 2^-64 collisions).  The oracle uses a private generator so that the
    global legacy stream - which the search consumes in program order - is untouched."""
    rs = np.random.RandomState(seed)
    n = (board_size + 2) ** 2
    hi = rs.randint(0, 2 ** 32, size=(4, n)).astype(np.uint64)
    lo = rs.randint(0, 2 ** 32, size=(4, n)).astype(np.uint64)
    return (hi << np.uint64(32)) | lo


class _Str:
    __slots__ = ("color", "stones", "libs")

    def __init__(self, color: int):
        self.color = color
        self.stones = set()
        self.libs = set()


class GoBoard:
    """go_board.py:17-129 (constructor / clear)."""

    def __init__(self, board_size: int, komi: float = 7.0, check_superko: bool = False):
        self.board_size = board_size
        self.width = board_size + 2
        self.komi = komi
        self.check_superko = check_superko
        w = self.width
        self.n_cells = w * w
        self.max_records = 3 * board_size * board_size          # board/constant.py:31
        self.onboard_pos = [x + y * w for y in range(1, board_size + 1)
                            for x in range(1, board_size + 1)]  # go_board.py:80-104 row-major
        self.zobrist = make_zobrist(board_size)
        self.clear()

    # -- helpers ---------------------------------------------------------------------
    def neighbor4(self, pos: int) -> List[int]:
        w = self.width
        return [pos - w, pos - 1, pos + 1, pos + w]            # go_board.py:44-54

    def cross4(self, pos: int) -> List[int]:
        w = self.width
        return [pos - w - 1, pos - w + 1, pos + w - 1, pos + w + 1]  # go_board.py:56-60

    def get_board_size(self) -> int:
        return self.board_size

    def clear(self):
        """go_board.py:109-129."""
        w = self.width
        self.board = [OUT_OF_BOARD] * self.n_cells
        for pos in self.onboard_pos:
            self.board[pos] = EMPTY
        self.sid = [0] * self.n_cells
        self.strings = {}
        self.moves = 1
        self.ko_move = 0
        self.ko_pos = 0
        self.prisoner = [0, 0]
        self.hash = 0
        self.rec_color = [EMPTY] * self.max_records
        self.rec_pos = [PASS] * self.max_records
        self.rec_hash = np.zeros(self.max_records, dtype=np.uint64)
        assert w * w == self.n_cells

    def copy_from(self, src: "GoBoard"):
        """copy_board, go_board.py:611-626 (check_superko is not copied there either)."""
        self.board = src.board[:]
        self.sid = src.sid[:]
        self.strings = {}
        for k, s in src.strings.items():
            t = _Str(s.color)
            t.stones = set(s.stones)
            t.libs = set(s.libs)
            self.strings[k] = t
        self.moves = src.moves
        self.ko_move = src.ko_move
        self.ko_pos = src.ko_pos
        self.prisoner = src.prisoner[:]
        self.hash = src.hash
        self.rec_color = src.rec_color[:]
        self.rec_pos = src.rec_pos[:]
        self.rec_hash = src.rec_hash.copy()

    def clone(self) -> "GoBoard":
        b = GoBoard(self.board_size, self.komi, self.check_superko)
        b.copy_from(self)
        return b

    # -- record (board/record.py) ----------------------------------------------------
    def _record(self, color: int, pos: int):
        """record.py:30-44: moves beyond MAX_RECORDS are silently dropped."""
        if self.moves < self.max_records:
            self.rec_color[self.moves] = color
            self.rec_pos[self.moves] = pos
            self.rec_hash[self.moves] = self.hash

    def record_pos(self, index: int) -> int:
        """record.py:65-74 (position component)."""
        return self.rec_pos[index]

    # -- string bookkeeping ----------------------------------------------------------
    def num_liberties(self, pos: int) -> int:
        """string.py:349-359; id 0 (empty / border) has 0 liberties."""
        sid = self.sid[pos]
        return len(self.strings[sid].libs) if sid else 0

    def string_size(self, pos: int) -> int:
        sid = self.sid[pos]
        return len(self.strings[sid].stones) if sid else 0

    def _new_id(self) -> int:
        """string.py:378-381: lowest free id starting at 1."""
        k = 1
        while k in self.strings:
            k += 1
        return k

    def _capture(self, sid: int) -> List[int]:
        """remove_string, string.py:292-325: clear the stones and give the vacated
        points back as liberties to every adjacent live string."""
        dead = self.strings.pop(sid)
        stones = sorted(dead.stones)
        for p in stones:
            self.board[p] = EMPTY
            self.sid[p] = 0
        for p in stones:
            for n in self.neighbor4(p):
                k = self.sid[n]
                if k:
                    self.strings[k].libs.add(p)
        return stones

    def put_stone(self, pos: int, color: int):
        """go_board.py:131-185.  No legality check, exactly like the reference."""
        if pos == PASS:
            self._record(color, pos)
            self.moves += 1
            return
        other = opponent(color)
        self.board[pos] = color
        self.hash ^= int(self.zobrist[color][pos])

        friends = []
        captured = 0
        for n in self.neighbor4(pos):
            c = self.board[n]
            if c == color:
                self.strings[self.sid[n]].libs.discard(pos)
                friends.append(self.sid[n])
            elif c == other:
                s = self.strings[self.sid[n]]
                s.libs.discard(pos)
                if not s.libs:
                    stones = self._capture(self.sid[n])
                    captured += len(stones)
                    for p in stones:
                        self.hash ^= int(self.zobrist[other][p])
        if color == BLACK:
            self.prisoner[0] += captured
        elif color == WHITE:
            self.prisoner[1] += captured

        empties = [n for n in self.neighbor4(pos) if self.board[n] == EMPTY]
        if not friends:
            k = self._new_id()
            s = _Str(color)
            s.stones.add(pos)
            s.libs.update(empties)
            self.strings[k] = s
            self.sid[pos] = k
            # ko rule, go_board.py:173-177: lone new stone that captured exactly one
            # stone and has exactly one liberty; ko point = smallest (= only) liberty.
            if captured == 1 and len(s.libs) == 1:
                self.ko_move = self.moves
                self.ko_pos = min(s.libs)
        else:
            ids = sorted(set(friends))                     # connect into the smallest id
            dst = self.strings[ids[0]]                     # string.py:453-458
            dst.stones.add(pos)
            self.sid[pos] = ids[0]
            dst.libs.update(empties)
            for k in ids[1:]:
                src = self.strings.pop(k)
                dst.stones |= src.stones
                dst.libs |= src.libs
                for p in src.stones:
                    self.sid[p] = ids[0]
            dst.libs.discard(pos)

        self._record(color, pos)
        self.moves += 1

    # -- legality ----------------------------------------------------------------------
    def n_empty_neighbors(self, pos: int) -> int:
        """pattern.py:21-30,142-151 (nb4_empty over pat3) == empty orthogonal cells."""
        return sum(1 for n in self.neighbor4(pos) if self.board[n] == EMPTY)

    def _is_suicide(self, pos: int, color: int) -> bool:
        """go_board.py:237-258."""
        other = opponent(color)
        for n in self.neighbor4(pos):
            c = self.board[n]
            if c == other and self.num_liberties(n) == 1:
                return False
            if c == color and self.num_liberties(n) > 1:
                return False
        return True

    def is_legal(self, pos: int, color: int) -> bool:
        """go_board.py:260-304."""
        if self.board[pos] != EMPTY:
            return False
        if self.n_empty_neighbors(pos) == 0 and self._is_suicide(pos, color):
            return False
        if self.ko_pos == pos and self.ko_move == self.moves - 1:
            return False
        if self.check_superko and pos != PASS:
            # go_board.py:285-301: hypothetically remove EVERY adjacent one-liberty
            # string (colour not checked; keys of the opponent colour are used for all of
            # them - reference quirk), add the stone, and look the hash up in the whole
            # fixed-size history array (unused slots hold 0).
            other = opponent(color)
            h = self.hash
            for k in set(self.sid[n] for n in self.neighbor4(pos)):
                if k and len(self.strings[k].libs) == 1:
                    for p in self.strings[k].stones:
                        h ^= int(self.zobrist[other][p])
            h ^= int(self.zobrist[color][pos])
            if np.any(self.rec_hash == np.uint64(h)):
                return False
        return True

    def get_all_legal_pos(self, color: int) -> List[int]:
        """go_board.py:400-409."""
        return [pos for pos in self.onboard_pos if self.is_legal(pos, color)]

    def check_self_atari_stone(self, pos: int, color: int) -> int:
        """go_board.py:327-365."""
        libs = set(n for n in self.neighbor4(pos) if self.board[n] == EMPTY)
        if len(libs) > 1:
            return 0
        other = opponent(color)
        seen = []
        size = 0
        for n in self.neighbor4(pos):
            c = self.board[n]
            if c == color:
                k = self.sid[n]
                if k in seen:
                    continue
                libs |= self.strings[k].libs
                if len(libs) >= 3:
                    return 0
                size += len(self.strings[k].stones)
                seen.append(k)
            elif c == other:
                if self.num_liberties(n) == 1:
                    return 0
        return size + 1

    def pat3(self, pos: int) -> int:
        """Neighbourhood code of ``pos`` (pattern.py:47-50,119-140)."""
        w = self.width
        offs = (-w - 1, -w, -w + 1, -1, 1, w - 1, w, w + 1)
        code = 0
        for k, d in enumerate(offs):
            code |= self.board[pos + d] << (2 * k)
        return code

    def eye_color(self, pos: int) -> int:
        """pattern.py:153-162."""
        return int(EYE_TABLE[self.pat3(pos)])

    def is_complete_eye(self, pos: int, color: int) -> bool:
        """go_board.py:367-397."""
        if self.eye_color(pos) != color:
            return False
        count = 0
        edge = False
        for c in self.cross4(pos):
            v = self.board[c]
            if v == color or v == OUT_OF_BOARD:
                count += 1
            elif v == EMPTY and self.eye_color(c) == color:
                count += 1
            if v == OUT_OF_BOARD:
                edge = True
        return (edge and count == 4) or (not edge and count >= 3)

    def search_candidates(self, color: int) -> List[int]:
        """Candidate filter of MCTSTree.expand_node, mcts/tree.py:260-264 (PASS last)."""
        out = [p for p in self.get_all_legal_pos(color)
               if self.check_self_atari_stone(p, color) < 7
               and not self.is_complete_eye(p, color)]
        out.append(PASS)
        return out

    # -- NN view -------------------------------------------------------------------------
    def get_board_data(self) -> List[int]:
        """go_board.py:468-478 with sym = 0."""
        return [self.board[p] for p in self.onboard_pos]

    def get_komi(self) -> float:
        return self.komi

    def count_score(self) -> int:
        """go_board.py:561-608, including its quirks: stones in atari count as dead;
        each empty point takes its colour from its DIRECT neighbours only (the flood fill
        re-queues the same coordinate, :592, so it never spreads); colourings are written
        into the working copy (:600-601 writes ``board[pos]``) and feed later points."""
        work = self.board[:]
        for pos in self.onboard_pos:
            if self.board[pos] in (BLACK, WHITE) and self.num_liberties(pos) == 1:
                work[pos] = EMPTY
        for pos in self.onboard_pos:
            if work[pos] != EMPTY:
                continue
            color = EMPTY
            for n in self.neighbor4(pos):
                v = work[n]
                if v in (BLACK, WHITE):
                    if color == EMPTY:
                        color = v
                    elif color != v:
                        color = OUT_OF_BOARD
            work[pos] = color
        return work.count(BLACK) - work.count(WHITE)
