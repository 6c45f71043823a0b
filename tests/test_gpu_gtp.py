"""GTP front end (next row 8(f).2) on the device search: a command script end to end; genmove
equals MCTSTree.search_best_move on the same position and random state.

This file compares the product WITH ITSELF (command loop vs direct search).  What pins the GTP row to the
reference are the lz- / cgos-analysis strings and PV lines (byte-identical to reference-recorded goldens,
tests/test_gpu_search.py) and the search itself (tree fixtures); the command loop has no reference recording."""
import io
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_script(client, script):
    old_out = sys.stdout
    sys.stdout = io.StringIO()
    try:
        client.stdin = io.StringIO(script)
        client.run()
        return sys.stdout.getvalue()
    finally:
        sys.stdout = old_out


def test_gtp_session_matches_direct_search(tmp_path):
    from oracle.stubnet import StubNet
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.board.coordinate import Coordinate
    from tamago_amd.gtp.client import GtpClient
    from tamago_amd.mcts.time_manager import TimeControl, TimeManager
    from tamago_amd.mcts.tree import MCTSTree
    from tests.helpers import load_json

    client = GtpClient(9, True, StubNet(8), visits=60, batch_size=16, tree_size=256,
                       mode=TimeControl.STRICT_PLAYOUT)
    np.random.seed(5)
    out = run_script(client, "1 protocol_version\nname\nknown_command genmove\nknown_command foo\n"
                             "boardsize 9\nclear_board\nkomi 6.5\nget_komi\nplay b E5\nplay w C3\n"
                             "7 genmove b\nundo\nbogus\nquit\n")
    blocks = out.split("\n\n")
    assert blocks[0] == "=1 2" and blocks[1] == "= TamaGo" and blocks[2] == "= true" and blocks[3] == "= false"
    assert blocks[7] == "= 6.5"
    assert blocks[10].startswith("=7 ")
    move = blocks[10][3:]
    assert blocks[11] == "= " and blocks[12] == "? unknown_command" and blocks[13] == "= "

    board = GoBoard(9, 6.5, True)
    coord = Coordinate(9)
    board.put_stone(coord.convert_from_gtp_format("E5"), 1)
    board.put_stone(coord.convert_from_gtp_format("C3"), 2)
    tree = MCTSTree(StubNet(8), tree_size=256, batch_size=16)
    np.random.seed(5)
    want = tree.search_best_move(board, 1, TimeManager(TimeControl.STRICT_PLAYOUT, 60), {})
    assert move == coord.convert_to_gtp_format(want)
    assert len(client.history) == 2                       # the generated move was undone

    # lz-genmove_analyze prints the analysis line, then "play <move>"; loadsgf replays a record
    np.random.seed(5)
    out = run_script(client, "lz-genmove_analyze b 0\nquit\n")
    assert out.startswith("= \ninfo move ") and f"\nplay {move}\n\n" in out
    games = load_json("selfplay_games.json")
    path = tmp_path / "g.sgf"
    path.write_text(games["1,16"], encoding="utf-8")
    out = run_script(client, f"loadsgf {path} 11\nlz-analyze w nonsense\nquit\n")
    assert out.split("\n\n")[0] == "= " and out.split("\n\n")[1].startswith("? lz-analyze")
    assert len(client.history) == 10 and client.board.moves == 11
