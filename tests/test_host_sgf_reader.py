"""SGF reader (next row 8(f).3) vs fields recorded from the reference's SGFReader on the
reference-recorded self-play games - CPU only."""
import hashlib

from tests.helpers import load_json


def test_reader_matches_reference_on_recorded_games_and_edge_cases():
    from tamago_amd.sgf.reader import SGFReader
    games = load_json("selfplay_games.json")
    meta = load_json("datagen_s9.json")["reader"]
    for key in sorted(games):
        want = meta[key]
        r = SGFReader(games[key], 9, literal=True)
        assert (r.size, r.komi, r.get_value_label(), r.get_n_moves()) == \
            (want["size"], want["komi"], want["value_label"], want["n_moves"])
        assert [r.get_move_data(i) for i in range(r.get_n_moves())] == want["moves"]
        assert list(r.get_moves()) == want["moves"]
        assert [int(r.get_color(i).value) for i in range(r.get_n_moves())] == want["colors"]
        assert [hashlib.sha256(r.get_comment(i).encode()).hexdigest()[:16]
                for i in range(r.get_n_moves())] == want["comment_sha"]
        assert r.get_comment(0) == want["comment0"]
        assert (r.application, r.black_player_name, r.white_player_name) == \
            (want["application"], want["black"], want["white"])
        assert r.get_move_data(r.get_n_moves()) == 0           # beyond the end: PASS
    edge = meta["edge"]
    r = SGFReader(edge["text"], 9, literal=True)
    assert r.komi == edge["komi"] and r.get_value_label() == edge["value_label"]
    assert [r.get_move_data(i) for i in range(r.get_n_moves())] == edge["moves"]
    assert [r.get_comment(i) for i in range(r.get_n_moves())] == edge["comments"]


def test_reader_reads_files(tmp_path):
    from tamago_amd.sgf.reader import SGFReader
    games = load_json("selfplay_games.json")
    path = tmp_path / "g.sgf"
    path.write_text(games["1,16"], encoding="utf-8")
    a = SGFReader(str(path), 9)
    b = SGFReader(games["1,16"], 9, literal=True)
    assert list(a.get_moves()) == list(b.get_moves()) and a.get_value_label() == b.get_value_label()
