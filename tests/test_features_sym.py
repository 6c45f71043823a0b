"""Training-side featurisation (next-tier row): all eight board symmetries and the
RL / SL policy targets vs reference-generated fixtures."""
import numpy as np
import pytest

from tests.helpers import load_json, load_npz, oracle_replay


@pytest.mark.parametrize("size", [9, 13, 19])
def test_oracle_symmetric_planes(size):
    from oracle.feature import generate_input_planes
    fix = load_npz(f"feat_s{size}.npz")
    brd = load_npz(f"board_s{size}.npz")
    for (g, ply, c, sym), want in zip(fix["sym_meta"], fix["sym_planes"]):
        board = oracle_replay(size, brd[f"g{g}_move"], brd[f"g{g}_color"], int(ply))
        got = generate_input_planes(board, int(c), int(sym))
        assert np.array_equal(got, want.astype(np.float32)), (g, ply, c, sym)


def _board_at(ply, moves):
    from tamago_amd.board.go_board import GoBoard
    board = GoBoard(9)
    for c, mv in moves[:ply]:
        pos = 0 if mv == "tt" else (ord(mv[0]) - 96) + (ord(mv[1]) - 96) * 11
        board.put_stone(pos, 1 if c == "B" else 2)
    return board


def test_policy_targets_host_logic():
    from tamago_amd.nn.feature import generate_rl_target_data, generate_target_data
    fix = load_json("rl_targets.json")
    for rec in fix["targets"]:
        board = _board_at(rec["ply"], fix["moves"])
        rl = generate_rl_target_data(board, rec["comment"], rec["sym"])
        assert [float(v).hex() for v in rl] == rec["rl"]
        sl = generate_target_data(board, rec["move"], rec["sym"])
        assert [int(v) for v in sl] == rec["sl"]


@pytest.mark.gpu
@pytest.mark.parametrize("size", [9, 13, 19])
def test_symmetric_featurize_kernel(size):
    from tamago_amd.nn.feature import featurize_batch, generate_input_planes
    from tamago_amd.board.go_board import GoBoard
    fix = load_npz(f"feat_s{size}.npz")
    brd = load_npz(f"board_s{size}.npz")
    cells, tm, prev, moves, sym, want = [], [], [], [], [], []
    for (g, ply, c, s_), planes in zip(fix["sym_meta"], fix["sym_planes"]):
        board = oracle_replay(size, brd[f"g{g}_move"], brd[f"g{g}_color"], int(ply))
        cells.append(board.get_board_data())
        tm.append(int(c))
        prev.append(board.record_pos(board.moves - 1))
        moves.append(board.moves)
        sym.append(int(s_))
        want.append(planes.astype(np.float32))
    out = featurize_batch(size, np.array(cells, dtype=np.uint8), np.array(tm), np.array(prev),
                          np.array(moves), np.array(sym))
    assert np.array_equal(out.cpu().numpy(), np.array(want))
    # single-position mirror of nn/feature.generate_input_planes
    board = GoBoard(size)
    board.put_stone(board.onboard_pos[size + 2], 1)
    one = generate_input_planes(board, 2, 5)
    assert one.shape == (6, size, size) and one[5].min() == -1.0 and one[2].sum() == 1.0
