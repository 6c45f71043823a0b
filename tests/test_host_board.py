"""Host-side product classes (GoBoard for the game loop, schedule, time manager) vs the
reference-generated fixtures - CPU only."""
import numpy as np
import pytest

from tests.helpers import load_json, load_npz


def _lst(arr):
    return [int(v) for v in arr if v >= 0]


@pytest.mark.parametrize("size,n_games", [(9, 6), (13, 2), (19, 2)])
def test_product_goboard_playouts(size, n_games):
    from tamago_amd.board.go_board import GoBoard, copy_board
    from tamago_amd.board.stone import Stone
    fix = load_npz(f"board_s{size}.npz")
    for g in range(n_games):
        superko = bool(fix[f"g{g}_superko"])
        board = GoBoard(size, 7.0, superko)
        moves, colors = fix[f"g{g}_move"], fix[f"g{g}_color"]
        step = 1 if size == 9 else 7
        for ply in range(len(moves)):
            board.put_stone(int(moves[ply]), Stone(int(colors[ply])))
            assert np.array_equal(np.array(board.get_board_data(), dtype=np.uint8),
                                  fix[f"g{g}_cells"][ply]), (g, ply)
            assert board.ko_pos == fix[f"g{g}_ko_pos"][ply]
            assert board.ko_move == fix[f"g{g}_ko_move"][ply]
            assert board.prisoner == list(fix[f"g{g}_pris"][ply])
            if ply % step == 0:
                assert board.get_all_legal_pos(Stone.BLACK) == _lst(fix[f"g{g}_legal_b"][ply]), (g, ply)
                assert board.get_all_legal_pos(2) == _lst(fix[f"g{g}_legal_w"][ply]), (g, ply)
                assert board.count_score() == fix[f"g{g}_score"][ply]
        other = GoBoard(size, 7.0, superko)
        copy_board(other, board)
        other.put_stone(0, 1)
        assert other.moves == board.moves + 1 and np.array_equal(other.cells, board.cells)


def test_schedule_and_coordinates():
    from tamago_amd.mcts.sequential_halving import get_candidates_and_visit_pairs
    from tamago_amd.board.coordinate import Coordinate
    for key, pairs in load_json("tables.json")["halving"].items():
        n0, v = (int(s) for s in key.split(","))
        assert [[k, c] for k, c in get_candidates_and_visit_pairs(n0, v).items()] == pairs
    c = Coordinate(9)
    assert c.convert_to_gtp_format(12) == "A9" and c.convert_from_gtp_format("A9") == 12
    assert c.convert_to_gtp_format(0) == "pass" and c.convert_to_sgf_format(12) == "aa"
    assert c.convert_from_gtp_format("J1") == 9 + 9 * 11
    assert c.convert_to_gtp_format(9 + 9 * 11) == "J1"


def test_time_manager_modes():
    from tamago_amd.mcts.time_manager import TimeManager, TimeControl
    from tamago_amd.mcts.node import MCTSNode
    root = MCTSNode(82)
    root.children_visits[:3] = [40, 10, 5]
    root.node_visits = 55
    tm = TimeManager(TimeControl.CONSTANT_PLAYOUT, constant_visits=80)
    assert tm.get_num_visits_threshold(1) == 80
    assert tm.is_move_decided(root, 80)            # 25 remaining < 30 lead
    assert not TimeManager(TimeControl.STRICT_PLAYOUT, 80).is_move_decided(root, 80)
    tm = TimeManager(TimeControl.CONSTANT_TIME, constant_time=2.0)
    assert tm.get_num_visits_threshold(1) == 40    # 20 visits/s default speed


def test_handicap_points_equal_the_reference_table():
    """board/handicap.py: the rule that produces the points against every entry (and every refusal) of the reference's table
    (tests/golden/handicap.json, written by tools/gen_golden_gtp.py from the imported reference)."""
    from tamago_amd.board.handicap import get_handicap_coordinates
    from tests.helpers import load_json
    table = load_json("handicap.json")
    assert len(table) == 17 * 12
    for key, want in table.items():
        size, n = (int(v) for v in key.split(","))
        assert get_handicap_coordinates(size, n) == want, key


def test_handicap_stones_are_stones_but_no_moves():
    from tamago_amd.board.go_board import GoBoard, copy_board
    from tamago_amd.board.stone import Stone
    board = GoBoard(9, 7.0, True)
    pos = [board.coordinate.convert_from_gtp_format(p) for p in ("G7", "C3")]
    for p in pos:
        board.put_handicap_stone(p, Stone.BLACK)
    assert board.moves == 1 and board.get_handicap_history() == pos
    data = board.get_board_data()
    assert sum(1 for v in data if v == 1) == 2 and sum(1 for v in data if v == 2) == 0
    assert not board.is_legal(pos[0], Stone.WHITE)
    other = GoBoard(9, 7.0, True)
    copy_board(other, board)
    assert other.get_handicap_history() == pos and other.get_board_data() == data
    board.clear()
    assert board.get_handicap_history() == [] and other.get_handicap_history() == pos
