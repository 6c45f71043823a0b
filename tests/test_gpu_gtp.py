"""GTP front end (next row 8(f).2) on the device search: a command script end to end; genmove
equals MCTSTree.search_best_move on the same position and random state.

The first test compares the product WITH ITSELF (command loop vs direct search); the last one replays a session the
REFERENCE's command loop recorded (tests/golden/gtp_session.json, tools/gen_golden_gtp.py) byte for byte.  The lz- /
cgos-analysis strings and PV lines are also pinned by reference-recorded goldens in tests/test_gpu_search.py."""
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
    assert blocks[0] == "=1 2" and blocks[1] == "= TamaGo" and blocks[2] == "= true" and blocks[3] == "? unknown command"
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


def test_gtp_session_equals_the_reference_recording():
    """The command loop against the REFERENCE's own: tools/gen_golden_gtp.py ran /root/reference/gtp/client.py on a scripted
    session (protocol commands, play / genmove / undo, an illegal move, handicap, time settings, lz- and cgos-genmove_analyze)
    with the deterministic stub network and recorded its stdout; the same script, seeds and network here must print the
    same bytes."""
    import random
    from oracle.stubnet import StubNet
    from tamago_amd.gtp.client import GtpClient
    from tamago_amd.mcts.time_manager import TimeControl
    from tests.helpers import load_json
    gold = load_json("gtp_session.json")
    client = GtpClient(9, True, StubNet(8), komi=7.0, visits=gold["visits"], batch_size=gold["batch_size"],
                       tree_size=gold["tree_size"], mode=TimeControl.STRICT_PLAYOUT)
    np.random.seed(gold["seed"])
    random.seed(gold["seed"])
    out = run_script(client, gold["script"])
    assert out == gold["stdout"]
