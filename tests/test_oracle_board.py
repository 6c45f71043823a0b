"""Oracle board engine vs reference-generated fixtures (tools/gen_golden.py)."""
import hashlib

import numpy as np
import pytest

from oracle.board import GoBoard, EYE_TABLE, BLACK, WHITE, PASS
from tests.helpers import load_npz, load_json


def test_eye_table():
    tables = load_json("tables.json")
    assert sorted(int(i) for i in np.nonzero(EYE_TABLE == BLACK)[0]) == tables["eye_black_codes"]
    assert sorted(int(i) for i in np.nonzero(EYE_TABLE == WHITE)[0]) == tables["eye_white_codes"]
    assert hashlib.sha256(EYE_TABLE.tobytes()).hexdigest() == tables["eye_sha256"]
    assert len(tables["eye_black_codes"]) == 90


def _cands(arr):
    return [int(v) for v in arr if v >= 0]


@pytest.mark.parametrize("size,n_games", [(9, 6), (13, 2), (19, 2)])
def test_playouts(size, n_games):
    fix = load_npz(f"board_s{size}.npz")
    for g in range(n_games):
        superko = bool(fix[f"g{g}_superko"])
        board = GoBoard(size, 7.0, superko)
        moves = fix[f"g{g}_move"]
        colors = fix[f"g{g}_color"]
        for ply in range(len(moves)):
            board.put_stone(int(moves[ply]), int(colors[ply]))
            cells = np.array(board.get_board_data(), dtype=np.uint8)
            assert np.array_equal(cells, fix[f"g{g}_cells"][ply]), (g, ply)
            libs = [board.num_liberties(p) for p in board.onboard_pos]
            assert np.array_equal(libs, fix[f"g{g}_libs"][ply]), (g, ply)
            sizes = [board.string_size(p) for p in board.onboard_pos]
            assert np.array_equal(sizes, fix[f"g{g}_sizes"][ply]), (g, ply)
            assert board.ko_pos == fix[f"g{g}_ko_pos"][ply]
            assert board.ko_move == fix[f"g{g}_ko_move"][ply]
            assert board.prisoner == list(fix[f"g{g}_pris"][ply])
            assert board.get_all_legal_pos(BLACK) == _cands(fix[f"g{g}_legal_b"][ply]), (g, ply)
            assert board.get_all_legal_pos(WHITE) == _cands(fix[f"g{g}_legal_w"][ply]), (g, ply)
            assert board.search_candidates(BLACK) == _cands(fix[f"g{g}_cand_b"][ply]), (g, ply)
            assert board.search_candidates(WHITE) == _cands(fix[f"g{g}_cand_w"][ply]), (g, ply)
            assert board.count_score() == fix[f"g{g}_score"][ply], (g, ply)


def test_copy_is_deep():
    a = GoBoard(9, check_superko=True)
    a.put_stone(a.onboard_pos[10], BLACK)
    b = a.clone()
    b.put_stone(b.onboard_pos[11], WHITE)
    assert a.moves == 2 and b.moves == 3
    assert a.board[a.onboard_pos[11]] == 0
    assert a.num_liberties(a.onboard_pos[10]) == 4 and b.num_liberties(b.onboard_pos[10]) == 3
    a.put_stone(PASS, WHITE)
    assert a.record_pos(2) == PASS and b.record_pos(2) == b.onboard_pos[11]
