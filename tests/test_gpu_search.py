"""Device-resident MCTS (tamago_amd/csrc/search.hip through the MCTSTree API) vs the
reference-generated tree fixtures and vs the oracle (needs a GPU).

The evaluator is oracle.stubnet.StubNet on all sides (bit-identical outputs on every
machine), so visit counts, float32-accumulated value sums, policies, chosen move, node
count, mini-batch sizes, a digest over the WHOLE tree and the position of numpy's global
RNG stream after the search must match exactly."""
import hashlib

import numpy as np
import pytest

from tests.helpers import load_json, load_npz, unhex

pytestmark = pytest.mark.gpu


def product_replay(size, moves, colors, upto, superko):
    from tamago_amd.board.go_board import GoBoard
    board = GoBoard(size, 7.0, superko)
    for mv, c in zip(moves[:upto], colors[:upto]):
        board.put_stone(int(mv), int(c))
    return board


def product_digest(tree, num_nodes):
    h = hashlib.sha256()
    for i in range(num_nodes):
        nd = tree.node[i]
        n = nd.num_children
        h.update(np.array(nd.action[:n], dtype=np.int32).tobytes())
        h.update(nd.children_index[:n].astype(np.int32).tobytes())
        h.update(nd.children_visits[:n].astype(np.int32).tobytes())
        h.update(nd.children_virtual_loss[:n].astype(np.int32).tobytes())
        h.update(nd.children_value_sum[:n].astype(np.float64).tobytes())
        h.update(nd.children_policy[:n].astype(np.float64).tobytes())
        h.update(np.array([nd.node_visits, nd.virtual_loss], dtype=np.int64).tobytes())
    return h.hexdigest()


def check_root(tree, mv, rec, digest=True):
    root = tree.get_root()
    n = root.num_children
    assert n == rec["n"]
    assert [int(a) for a in root.action[:n]] == rec["action"]
    assert [int(v) for v in root.children_visits[:n]] == rec["child_visits"]
    assert np.array_equal(root.children_value_sum[:n], unhex(rec["value_sum"]))
    assert np.array_equal(root.children_policy[:n], unhex(rec["policy"]))
    assert int(mv) == rec["move"]
    assert tree.num_nodes == rec["num_nodes"]
    assert int(root.node_visits) == rec["node_visits"]
    assert float(root.node_value_sum) == float.fromhex(rec["node_value_sum"])
    assert float(root.raw_value) == float.fromhex(rec["raw_value"])
    if digest:
        assert product_digest(tree, tree.num_nodes) == rec["digest"]


@pytest.mark.parametrize("size", [9, 13, 19])
def test_puct_vs_reference_golden(size):
    from oracle.stubnet import StubNet
    from tamago_amd.mcts.tree import MCTSTree
    from tamago_amd.mcts.time_manager import TimeManager, TimeControl
    brd = load_npz(f"board_s{size}.npz")
    for rec in [r for r in load_json(f"trees_s{size}.json") if r["kind"] == "puct"]:
        board = product_replay(size, brd["g0_move"], brd["g0_color"], rec["ply"], rec["superko"])
        cells_before = board.cells.copy()
        net = StubNet(salt=rec["seed"])
        tree = MCTSTree(net, tree_size=2048, batch_size=rec["batch"], cgos_mode=rec["cgos"])
        mode = TimeControl.STRICT_PLAYOUT if rec["mode"] == "STRICT" else TimeControl.CONSTANT_PLAYOUT
        np.random.seed(rec["seed"])
        mv = tree.search_best_move(board, rec["color"], TimeManager(mode, rec["visits"]), {})
        assert net.calls == rec["batches"], rec
        check_root(tree, mv, rec)
        assert float(np.random.random_sample()) == float.fromhex(rec["rng_after"])
        assert np.array_equal(board.cells, cells_before) and board.moves == rec["ply"] + 1
        # analysis strings of the GTP front end (lz-analyze / cgos-analyze), incl. the PVs
        root = tree.get_root()
        assert root.get_analysis(board, "lz", tree.get_pv_lists) == rec["analysis_lz"]
        assert root.get_analysis(board, "cgos", tree.get_pv_lists) == rec["analysis_cgos"]


@pytest.mark.parametrize("size", [9, 13, 19])
def test_gumbel_vs_reference_golden(size):
    from oracle.stubnet import StubNet
    from tamago_amd.mcts.tree import MCTSTree
    from tamago_amd.mcts.time_manager import TimeManager, TimeControl
    brd = load_npz(f"board_s{size}.npz")
    for rec in [r for r in load_json(f"trees_s{size}.json") if r["kind"] == "gumbel"]:
        board = product_replay(size, brd["g0_move"], brd["g0_color"], rec["ply"], rec["superko"])
        net = StubNet(salt=100 + rec["seed"])
        tree = MCTSTree(net, tree_size=160 if rec["visits"] <= 100 else 2048)
        np.random.seed(rec["seed"])
        mv = tree.generate_move_with_sequential_halving(
            board, rec["color"], TimeManager(TimeControl.CONSTANT_PLAYOUT, rec["visits"]), True)
        assert net.calls == rec["batches"], rec
        check_root(tree, mv, rec)
        root = tree.get_root()
        assert np.array_equal(root.noise, unhex(rec["noise"]))
        assert np.array_equal(root.calculate_improved_policy(), unhex(rec["improved"]))
        assert float(np.random.random_sample()) == float.fromhex(rec["rng_after"])


def test_device_play_matches_host_board():
    """tg_search_play (root boards resident on the device) vs the reference play-outs."""
    from oracle.stubnet import StubNet
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.mcts.engine import SearchEngine, HostEvaluator
    import torch
    size = 9
    fix = load_npz(f"board_s{size}.npz")
    games = [0, 1, 2, 3]
    eng = SearchEngine(size, len(games), 64, 4, HostEvaluator(StubNet(0), torch.device("cuda:0")),
                       check_superko=True)
    for t in range(len(games)):
        eng.set_root(t, GoBoard(size, 7.0, True), 1, np.random.RandomState(t).get_state())
    eng.root_eval(False)
    for ply in range(150):
        moves = np.array([int(fix[f"g{g}_move"][ply]) for g in games], dtype=np.int32)
        eng.play(moves)
        if ply % 10 == 9 or ply == 149:
            cells, mv, to_move = eng.read_positions()
            for t, g in enumerate(games):
                onboard = cells[t].reshape(size + 2, size + 2)[1:-1, 1:-1].reshape(-1)
                assert np.array_equal(onboard, fix[f"g{g}_cells"][ply]), (g, ply)
                assert mv[t] == ply + 2 and to_move[t] == 3 - int(fix[f"g{g}_color"][ply])
    # a search from the device-resident position equals a search from the host position
    ref_board = product_replay(size, fix["g0_move"], fix["g0_color"], 150, True)
    eng2 = SearchEngine(size, 1, 256, 16, HostEvaluator(StubNet(9), torch.device("cuda:0")),
                        check_superko=True)
    eng2.set_root(0, ref_board, int(to_move[0]), np.random.RandomState(77).get_state())
    eng2.root_eval(False)
    eng2.puct_batch(16)
    eng2.puct_batch(16)
    want = eng2.read_node(0, 0)
    eng.evaluator = HostEvaluator(StubNet(9), torch.device("cuda:0"))
    for t in range(len(games)):
        eng.set_stream(t, np.random.RandomState(77).get_state())
    eng.root_eval(False)
    eng.puct_batch(4)
    got = eng.read_node(0, 0)
    assert got.num_children == want.num_children and got.action == want.action


def test_edge_cases_pass_only_root_and_errors():
    """Root with PASS as the only candidate returns PASS after the root evaluation
    (tree.py:76-77); pool overflow and bad arguments are reported, not ignored."""
    import torch
    from oracle.stubnet import StubNet
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.lib import TamagoHipError
    from tamago_amd.mcts.engine import SearchEngine, HostEvaluator
    from tamago_amd.mcts.tree import MCTSTree
    from tamago_amd.mcts.time_manager import TimeManager, TimeControl
    # a full board of black stones with two eyes: white has no legal move, black only eye fills
    board = GoBoard(9)
    for pos in board.onboard_pos:
        if pos not in (12, 14):                     # A9 and C9 stay empty (two black eyes)
            board.cells[pos] = 1
    board.moves = 5
    tree = MCTSTree(StubNet(1), tree_size=64, batch_size=4)
    np.random.seed(0)
    assert tree.search_best_move(board, 2, TimeManager(TimeControl.STRICT_PLAYOUT, 20), {}) == 0
    root = tree.get_root()
    assert root.num_children == 1 and root.action[0] == 0 and tree.num_nodes == 1
    np.random.seed(0)
    assert tree.search_best_move(board, 1, TimeManager(TimeControl.STRICT_PLAYOUT, 20), {}) == 0
    assert tree.get_root().num_children == 1        # both eyes are complete eyes: PASS only
    # node pool too small for the visit budget -> TG_ERR_OVERFLOW surfaces at the next read
    small = SearchEngine(9, 1, 8, 16, HostEvaluator(StubNet(0), torch.device("cuda:0")))
    small.set_root(0, GoBoard(9), 1, np.random.RandomState(0).get_state())
    small.root_eval(False)
    small.puct_batch(16)
    with pytest.raises(TamagoHipError, match="node pool full"):
        small.read_node(0, 0)
    with pytest.raises(TamagoHipError):
        small.puct_batch(17)                         # more leaves than batch_size
    # MCTSTree grows the pool like the reference's "Tree is full" path and gives the same answer
    grow = MCTSTree(StubNet(2), tree_size=16, batch_size=8)
    ref = MCTSTree(StubNet(2), tree_size=256, batch_size=8)
    np.random.seed(4)
    a = grow.search_best_move(GoBoard(9), 1, TimeManager(TimeControl.STRICT_PLAYOUT, 100), {})
    np.random.seed(4)
    b = ref.search_best_move(GoBoard(9), 1, TimeManager(TimeControl.STRICT_PLAYOUT, 100), {})
    assert a == b and grow.tree_size >= 128
    assert np.array_equal(grow.get_root().children_visits, ref.get_root().children_visits)


def test_gumbel_packed_leaf_layout_equals_strided():
    """slots_per_tree = 0 (leaves of the trees back to back) gives the same trees as the strided
    layout, with ragged per-tree phases: an idle tree, a single-candidate tree (1 x 40 levels)
    and ordinary (8 x 3), (4 x 5) phases."""
    import torch
    from oracle.stubnet import StubNet
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.mcts.engine import SearchEngine, HostEvaluator

    def run(packed):
        eng = SearchEngine(9, 4, 160, 48, HostEvaluator(StubNet(3), torch.device("cuda:0")))
        boards = [GoBoard(9) for _ in range(4)]
        boards[1].put_stone(boards[1].onboard_pos[40], 1)
        boards[3].put_stone(boards[3].onboard_pos[10], 1)
        for t, b in enumerate(boards):
            eng.set_root(t, b, 1 if b.moves % 2 == 0 else 2, np.random.RandomState(10 + t).get_state())
        eng.root_eval(use_logit=True)
        eng.set_gumbel_noise()
        eng.gumbel_phase([8, 0, 1, 4], [3, 0, 40, 5], packed=packed)
        eng.gumbel_phase([4, 8, 0, 2], [5, 3, 0, 9], packed=packed)
        stats = eng.read_root_stats()
        nodes = eng.num_nodes()
        eng.close()
        return stats, nodes

    a, na = run(True)
    b, nb = run(False)
    assert np.array_equal(na, nb)
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    assert a["children_visits"][0].sum() == 8 * 3 + 4 * 5 and a["children_visits"][2].sum() == 40


@pytest.mark.parametrize("one_by_one", [False, True])
def test_gumbel_phase_reports_a_full_node_pool(one_by_one, monkeypatch):
    """A halving phase whose entries need more nodes than the pool has: the nodes that fit are handed out in entry order, the
    launch reports "node pool full" (tree.py:418-420's "Tree is full" path is the caller's: grow and search again) - in the usual
    mode (node numbers handed out by lanes) and one by one through the job ring (TG_GUMBEL_ONE_BY_ONE); a pool that fits gives
    the same tree in both modes."""
    import torch
    from oracle.stubnet import StubNet
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.lib import TamagoHipError
    from tamago_amd.mcts.engine import SearchEngine, HostEvaluator
    if one_by_one:
        monkeypatch.setenv("TG_GUMBEL_ONE_BY_ONE", "1")
    else:
        monkeypatch.delenv("TG_GUMBEL_ONE_BY_ONE", raising=False)

    def run(tree_size):
        eng = SearchEngine(9, 2, tree_size, 48, HostEvaluator(StubNet(6), torch.device("cuda:0")))
        for t in range(2):
            eng.set_root(t, GoBoard(9), 1, np.random.RandomState(70 + t).get_state())
        eng.root_eval(use_logit=True)
        eng.set_gumbel_noise()
        eng.gumbel_phase([16, 8], [1, 2])
        eng.gumbel_phase([8, 8], [3, 4])
        stats = eng.read_root_stats()
        nodes = eng.num_nodes()
        eng.close()
        return stats, nodes

    stats, nodes = run(64)
    assert nodes[0] == 1 + 8 and nodes[1] == 1 + 8           # a root child's node is made by the first descent that finds it visited
    assert stats["children_visits"][0].sum() == 16 + 24 and stats["children_visits"][1].sum() == 16 + 32
    if one_by_one:
        monkeypatch.delenv("TG_GUMBEL_ONE_BY_ONE")
        usual, usual_nodes = run(64)
        assert np.array_equal(nodes, usual_nodes)
        for key in stats:
            assert np.array_equal(stats[key], usual[key]), key
        monkeypatch.setenv("TG_GUMBEL_ONE_BY_ONE", "1")
    with pytest.raises(TamagoHipError, match="node pool full"):
        run(6)                                                   # either tree wants 9 nodes


def test_library_streams_equal_host_streams():
    """The library-owned legacy streams (tg_search_seed_stream / feed / advance / draw_noise) and
    the numpy-side feed (ExpStream + tg_search_set_rng / rng_consumed / set_noise) give the same
    trees, the same Gumbel noise and the same generator state afterwards."""
    import torch
    from oracle.stubnet import StubNet
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.mcts.engine import SearchEngine, HostEvaluator

    def run(host_streams):
        eng = SearchEngine(9, 3, 200, 32, HostEvaluator(StubNet(5), torch.device("cuda:0")),
                           host_streams=host_streams)
        for t in range(3):
            eng.set_root(t, GoBoard(9), 1, np.random.RandomState(40 + t).get_state())
        eng.root_eval(False, first_batch=7)
        eng.puct_batch(7)
        eng.prefetch_rng(32)
        eng.puct_batch(32)
        eng.puct_batch(5)
        puct = eng.read_root_stats()
        for t in range(3):
            eng.set_root(t, GoBoard(9), 1)                 # new search, streams continue
        eng.root_eval(use_logit=True)
        noise = eng.set_gumbel_noise().copy()
        eng.gumbel_phase([16, 8, 4], [2, 4, 8])
        eng.gumbel_phase([8, 4, 2], [4, 8, 16])
        gum = eng.read_root_stats()
        states = [eng.streams[t].final_state() for t in range(3)]
        eng.close()
        return puct, noise, gum, states

    a = run(False)
    b = run(True)
    for key in a[0]:
        assert np.array_equal(a[0][key], b[0][key]), key
        assert np.array_equal(a[2][key], b[2][key]), key
    assert np.array_equal(a[1], b[1])
    for sa, sb in zip(a[3], b[3]):
        ga, gb = np.random.RandomState(), np.random.RandomState()
        ga.set_state(sa)
        gb.set_state(sb)
        assert np.array_equal(ga.random_sample(8), gb.random_sample(8))


def test_search_with_callback_and_ponder():
    """search_with_callback (tree.py:177-196): the per-descent (node, child) paths equal the
    oracle's search_mcts paths at batch size 1; ponder (tree.py:108-127) with input already
    waiting on stdin stops after one descent and prints the analysis line."""
    import io
    import os
    import sys
    from oracle.board import GoBoard as OBoard
    from oracle.stubnet import StubNet
    from oracle.tree import MCTSTree as OTree
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.mcts.tree import MCTSTree

    paths = []

    def callback(path):
        paths.append(list(path))
        return len(paths) == 40

    tree = MCTSTree(StubNet(6), tree_size=128, batch_size=8)
    np.random.seed(21)
    tree.search_with_callback(GoBoard(9), 1, callback)
    after = np.random.random_sample(3)

    oracle = OTree(StubNet(6), 9, tree_size=128, batch_size=1)
    np.random.seed(21)
    board = OBoard(9)
    oracle._initialize_search(board, 1)
    want = []
    scratch = board.clone()
    for _ in range(40):
        path = []
        scratch.copy_from(board)
        oracle.search_mcts(scratch, 1, oracle.current_root, path)
        want.append([(int(a), int(b)) for a, b in path])
    assert paths == want
    assert np.array_equal(after, np.random.random_sample(3))           # same stream position
    assert tree.num_nodes == oracle.num_nodes

    # ponder: stdin readable from the start -> one descent, then the lz analysis line
    rd, wr = os.pipe()
    os.write(wr, b"stop\n")
    old_in, old_out = sys.stdin, sys.stdout
    sys.stdin, sys.stdout = os.fdopen(rd, "r"), io.StringIO()
    try:
        pond = MCTSTree(StubNet(6), tree_size=128, batch_size=8)
        np.random.seed(21)
        pond.ponder(GoBoard(9), 1, {"mode": "lz", "interval": 0, "ponder": True})
        text = sys.stdout.getvalue()
    finally:
        sys.stdin.close()
        os.close(wr)
        sys.stdin, sys.stdout = old_in, old_out
    root = pond.get_root()
    assert root.node_visits == 1 and pond.num_nodes == 2                # root + the one descent that expanded a child
    assert text.startswith("info move ") and text.endswith("\n")


@pytest.mark.parametrize("cfg", ["9 24 64 4", "9 1 256 4", "19 6 32 4", "9 16 64 5", "13 6 48 4"])
def test_pipelined_selection_equals_serial(cfg):
    """The pipelined PUCT selection kernels - select_puct_mpipe_kernel (descents pipelined over several selector
    waves + board workers, the default up to 256 trees) and select_puct_pipe_kernel (selector + two workers per
    tree, for more trees; forced here with TG_SELECT_MPIPE_TREES=0) - build exactly the trees of the one-wavefront
    kernel (TG_SELECT_SERIAL=1): ragged roots, superko, several mini-batches with a short last one.  (13x13 has the two
    pipelined kernels, no split instantiation: the "split" variants run the default there.)"""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_select_digest.py")
    outs = []
    for variant in ("serial", "mpipe", "mpipe", "pipe", "split", "split", "split1"):
        env = dict(os.environ)
        env.pop("TG_SELECT_SERIAL", None)
        env.pop("TG_SELECT_MPIPE_TREES", None)
        env.pop("TG_SPLIT_CFG", None)
        # select_puct_split_kernel (<= 16 trees): one selecting workgroup + two of workers per tree; "split1": one of workers
        env["TG_SELECT_SPLIT"] = "1" if variant.startswith("split") else "0"
        if variant == "split1":
            env["TG_SPLIT_CFG"] = "11016" if cfg.startswith("9 ") else "11007"
        if variant == "serial":
            env["TG_SELECT_SERIAL"] = "1"
        elif variant == "pipe":
            env["TG_SELECT_MPIPE_TREES"] = "0"
        res = subprocess.run([sys.executable, script] + cfg.split(), env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append(res.stdout.strip().splitlines()[-1])
    assert len(set(outs)) == 1, outs


@pytest.mark.gpu
@pytest.mark.parametrize("size,trees,batches", [(9, 1, [256, 256, 256, 232]), (9, 3, [32, 32, 7]), (19, 1, [64] * 5 + [13])])
def test_chained_mini_batches_equal_the_per_mini_batch_loop(size, trees, batches):
    """tg_search_puct_chain (all mini-batches of a STRICT_PLAYOUT search queued in one call, the random window sent in pieces) against
    puct_batch called once per mini-batch, on freshly seeded streams as search_best_move starts them: same root statistics, same
    node counts, same generator state afterwards - and MCTSTree.search takes the chained path exactly when nothing between the
    mini-batches depends on their results."""
    import torch
    from oracle.net import make_state_dict
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.mcts.engine import SearchEngine, DeviceEvaluator
    from tamago_amd.nn.network.dual_net import DualNet
    net = DualNet(torch.device("cuda:0"), size)
    net.load_state_dict(make_state_dict(size, 3, 1.3))

    def run(chain):
        eng = SearchEngine(size, trees, sum(batches) + 64, max(batches), DeviceEvaluator(net))
        for t in range(trees):
            eng.set_root(t, GoBoard(size), 1, np.random.RandomState(70 + t).get_state())
        eng.root_eval(False)
        assert eng.can_chain(sum(batches))
        if chain:
            eng.puct_chain(batches)
        else:
            for k in batches:
                eng.puct_batch(k)
        stats = eng.read_root_stats()
        nodes = eng.num_nodes().copy()
        states = [eng.streams[t].final_state() for t in range(trees)]
        sizes = list(eng.evaluator.batches)
        eng.close()
        return stats, nodes, states, sizes

    a, b = run(False), run(True)
    for key in a[0]:
        assert np.array_equal(a[0][key], b[0][key]), key
    assert np.array_equal(a[1], b[1]) and a[3] == b[3]
    for sa, sb in zip(a[2], b[2]):
        ga, gb = np.random.RandomState(), np.random.RandomState()
        ga.set_state(sa)
        gb.set_state(sb)
        assert np.array_equal(ga.random_sample(8), gb.random_sample(8))


@pytest.mark.gpu
@pytest.mark.parametrize("size,trees", [(9, 3), (19, 1)])
def test_play_of_the_most_visited_child_on_the_device_equals_the_host_choice(size, trees):
    """tg_search_play with -2 (the play kernel takes np.argmax(children_visits[:num_children]) itself, node.py:167-175) leaves the
    same root positions as reading the roots back, choosing on the host and playing that move."""
    import torch
    from oracle.net import make_state_dict
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.mcts.engine import SearchEngine, DeviceEvaluator
    from tamago_amd.nn.network.dual_net import DualNet
    net = DualNet(torch.device("cuda:0"), size)
    net.load_state_dict(make_state_dict(size, 5, 1.3))

    def run(on_device):
        eng = SearchEngine(size, trees, 200, 32, DeviceEvaluator(net))
        for t in range(trees):
            eng.set_root(t, GoBoard(size), 1, np.random.RandomState(11 + t).get_state())
        out = []
        for _ in range(3):
            eng.root_eval(False)
            for _ in range(3):
                eng.puct_batch(32)
            if on_device:
                eng.play(np.full(trees, -2, dtype=np.int32))
            else:
                nc, action, visits = eng.read_roots()
                masked = np.where(np.arange(eng.A)[None, :] < nc[:, None], visits, -1)
                eng.play(action[np.arange(trees), np.argmax(masked, axis=1)].astype(np.int32))
            out.append(eng.read_positions())
        eng.close()
        return out

    for (ca, ma, ta), (cb, mb, tb) in zip(run(False), run(True)):
        assert np.array_equal(ca, cb) and np.array_equal(ma, mb) and np.array_equal(ta, tb)


def test_split_selector_with_a_muted_worker_workgroup_reports_a_stall_instead_of_hanging():
    """select_puct_split_kernel spans three workgroups per tree that hand work to each other through memory (tagged job
    entries, agent scope) - correct only while all of them are resident.  Every wait is bounded: with the worker workgroups
    muted (TG_SPLIT_TEST_MUTE: they exit at once, as if they had never been scheduled) the selecting half runs into its limits,
    the launch ENDS, and the next read reports a stalled pipeline for the tree - the library's answer to a device that cannot
    hold a tree's workgroups together (a shared device takes the one-workgroup kernels from the start, TG_SHARED_DEVICE).  A
    fresh search in the same process, hook off, is unaffected.  (Subprocess: the hook is read once per launch from the
    environment, the error is sticky per tree.)"""
    import os
    import subprocess
    import sys
    code = r"""
import sys, time, numpy as np, torch
sys.path.insert(0, %r)
from oracle.stubnet import StubNet
from tamago_amd.board.go_board import GoBoard
from tamago_amd.mcts.engine import SearchEngine, HostEvaluator
from tamago_amd.lib import TamagoHipError
dev = torch.device("cuda", 0)
eng = SearchEngine(9, 2, 600, 64, HostEvaluator(StubNet(salt=5), dev))
board = GoBoard(9, 7.0, False)
for t in range(2):
    eng.set_root(t, board, 1, np.random.RandomState(t).get_state())
eng.root_eval(False, first_batch=64)
t0 = time.time()
try:
    eng.puct_batch(64)
    eng.read_root_stats()
    print("NO ERROR")
except TamagoHipError as exc:
    print("ERROR after %%.1f s: %%s" %% (time.time() - t0, exc))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TG_DEBUG_KNOBS="1", TG_SPLIT_TEST_MUTE="1", TG_SELECT_SPLIT="1")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    last = res.stdout.strip().splitlines()[-1]
    assert last.startswith("ERROR after") and "pipeline stalled" in last, res.stdout[-1000:] + res.stderr[-1000:]
    env.pop("TG_SPLIT_TEST_MUTE")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.strip().splitlines()[-1] == "NO ERROR", res.stdout[-1000:] + res.stderr[-1000:]
