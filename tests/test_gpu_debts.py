"""Control branches and API corners of the hot path that the golden-tree tests do not reach:
RESIGN, time-based budgets, in-place pool growth (search and ponder), the BatchQueue view of the
device leaf queue, load_network, board-size mismatches, concurrent 19x19 forwards."""
import io
import os
import sys
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class SideNet:
    """Evaluator whose value head only looks at the side-to-move plane (plane 5: +1 black / -1
    white): value distribution `black` for black-to-move positions, `white` otherwise, flat policy.
    With (black, white) = (win, loss) every backed-up root edge value is 0 -> RESIGN; mirrored, 1."""

    def __init__(self, black, white, size=9):
        self.black = torch.tensor(black, dtype=torch.float32)
        self.white = torch.tensor(white, dtype=torch.float32)
        self.a = size * size + 1

    def _value(self, planes):
        black_to_move = (planes[:, 5, 0, 0] > 0).unsqueeze(1)
        return torch.where(black_to_move, self.black.unsqueeze(0), self.white.unsqueeze(0)).contiguous()

    def inference(self, planes):
        return torch.full((planes.shape[0], self.a), 1.0 / self.a), self._value(planes)

    def inference_with_policy_logits(self, planes):
        return torch.zeros((planes.shape[0], self.a)), self._value(planes)


def test_resign_branch_matches_oracle():
    """tree.py:100-103: best child's mean value < RESIGN_THRESHOLD -> RESIGN (-1); the mirrored
    evaluator plays a move.  Same answers and visit counts as the CPU oracle."""
    from oracle.board import GoBoard as OBoard
    from oracle.tree import MCTSTree as OTree, TimeManager as OTM, TimeControl as OTC
    from tamago_amd.board.constant import RESIGN
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.mcts.time_manager import TimeManager, TimeControl
    from tamago_amd.mcts.tree import MCTSTree
    win, loss = [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]
    got = []
    for black, white in ((win, loss), (loss, win)):
        tree = MCTSTree(SideNet(black, white), tree_size=256, batch_size=8)
        np.random.seed(9)
        mv = tree.search_best_move(GoBoard(9), 1, TimeManager(TimeControl.STRICT_PLAYOUT, 120), {})
        otree = OTree(SideNet(black, white), 9, tree_size=256, batch_size=8)
        np.random.seed(9)
        omv = otree.search_best_move(OBoard(9), 1, OTM(OTC.STRICT_PLAYOUT, 120))
        assert mv == omv
        n = otree.get_root().num_children
        assert np.array_equal(tree.get_root().children_visits[:n], otree.get_root().children_visits[:n])
        assert np.array_equal(tree.get_root().children_value_sum[:n], otree.get_root().children_value_sum[:n])
        got.append(mv)
    assert got[0] == RESIGN and got[1] > 0
    # Gumbel path: never_resign suppresses it (tree.py:351-354)
    tree = MCTSTree(SideNet(win, loss), tree_size=256, batch_size=8)
    np.random.seed(9)
    tm = TimeManager(TimeControl.CONSTANT_PLAYOUT, 32)
    assert tree.generate_move_with_sequential_halving(GoBoard(9), 1, tm, False) == RESIGN
    np.random.seed(9)
    assert tree.generate_move_with_sequential_halving(GoBoard(9), 1, tm, True) >= 0


def test_time_budgets_through_search_best_move():
    """CONSTANT_TIME and TIME_CONTROL (time_manager.py:61-83): the visit threshold is
    search_speed x time limit, the search ends at the threshold or when the clock runs out, the
    player's remaining time is charged, the measured speed feeds the next move's threshold."""
    import time
    from oracle.stubnet import StubNet
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.mcts.constant import VISITS_PER_SEC
    from tamago_amd.mcts.time_manager import TimeManager, TimeControl
    from tamago_amd.mcts.tree import MCTSTree
    tree = MCTSTree(StubNet(4), tree_size=1024, batch_size=16)
    # CONSTANT_TIME: threshold = VISITS_PER_SEC * 0.2 on the first move
    tm = TimeManager(TimeControl.CONSTANT_TIME, constant_time=0.2)
    want = max(int(VISITS_PER_SEC * 0.2), 1)
    np.random.seed(1)
    t0 = time.time()
    mv = tree.search_best_move(GoBoard(9), 1, tm, {})
    spent = time.time() - t0
    root = tree.get_root()
    assert mv > 0 and 1 <= root.node_visits <= want
    assert spent < 0.2 + 2.0                                  # stops by count or by clock
    assert tm.search_speed > 0 and tm.search_speed != VISITS_PER_SEC      # measured on this move
    assert tm.get_num_visits_threshold(1) == max(int(tm.search_speed * 0.2), 1)
    # a budget the clock cannot cover: is_time_over() ends the search between mini-batches
    slow = TimeManager(TimeControl.CONSTANT_TIME, constant_time=0.05)
    slow.search_speed = 1e6                                   # threshold 50 000 visits
    np.random.seed(1)
    t0 = time.time()
    tree.search_best_move(GoBoard(9), 1, slow, {})
    assert time.time() - t0 < 3.0 and tree.get_root().node_visits < 50000
    # TIME_CONTROL: a tenth of the mover's clock, charged afterwards
    tc = TimeManager(TimeControl.TIME_CONTROL, remaining_time=2.0)
    tc.initialize()
    np.random.seed(1)
    tree.search_best_move(GoBoard(9), 2, tc, {})
    assert tc.time_limit == pytest.approx(0.2)
    assert tc.remaining_time[0] == 2.0 and 1.0 < tc.remaining_time[1] < 2.0


def test_pool_grows_in_place_like_the_reference(capsys):
    """tree.py:254-258: the node list doubles when it fills up and the search goes on.  A search
    that starts with 16 nodes gives the tree a 512-node search gives, the clock is not reset, and
    ponder keeps growing instead of stopping."""
    from oracle.stubnet import StubNet
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.mcts.time_manager import TimeManager, TimeControl
    from tamago_amd.mcts.tree import MCTSTree
    grow = MCTSTree(StubNet(2), tree_size=16, batch_size=8)
    ref = MCTSTree(StubNet(2), tree_size=512, batch_size=8)
    np.random.seed(4)
    a = grow.search_best_move(GoBoard(9), 1, TimeManager(TimeControl.STRICT_PLAYOUT, 300), {})
    np.random.seed(4)
    b = ref.search_best_move(GoBoard(9), 1, TimeManager(TimeControl.STRICT_PLAYOUT, 300), {})
    assert a == b and grow.tree_size in (256, 512) and grow.num_nodes == ref.num_nodes
    ra, rb = grow.get_root(), ref.get_root()
    assert np.array_equal(ra.children_visits, rb.children_visits)
    assert np.array_equal(ra.children_value_sum, rb.children_value_sum)
    assert "Tree is full. Allocate new space 16 -> 32" in capsys.readouterr().err
    # deep nodes survived the copy: the principal variation of both trees is the same
    assert grow.get_best_move_sequence([], 0) == ref.get_best_move_sequence([], 0)
    # ponder without input: grows 32 -> 64 -> 128, stops where the cap forbids the next doubling
    pond = MCTSTree(StubNet(2), tree_size=32, batch_size=8)
    pond.ponder_max_nodes = 128
    np.random.seed(4)
    pond.ponder(GoBoard(9), 1, {"mode": "lz", "interval": 100, "ponder": False})
    visits = pond.get_root().node_visits
    assert pond.tree_size == 128 and 96 <= visits <= 128 and visits % 8 == 0
    same = MCTSTree(StubNet(2), tree_size=512, batch_size=8)
    np.random.seed(4)
    same.search_best_move(GoBoard(9), 1, TimeManager(TimeControl.STRICT_PLAYOUT, visits), {})
    assert np.array_equal(pond.get_root().children_visits, same.get_root().children_visits)


def test_multi_tree_pool_growth_is_transactional_and_keeps_every_tree():
    """tg_search_grow on a 6-tree engine: every tree's rows survive the re-strided copy - a digest over
    root statistics and sampled nodes of all trees equals that of an engine created at the final size -, and a growth
    whose allocation fails (absurd size) returns an error and leaves the handle exactly as it was."""
    import hashlib
    import torch
    from oracle.stubnet import StubNet
    from tamago_amd import lib as tl
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.mcts.engine import SearchEngine, HostEvaluator

    def run(tree_size, grow_to=None, fail=False):
        eng = SearchEngine(9, 6, tree_size, 8, HostEvaluator(StubNet(5), torch.device("cuda:0")), check_superko=True)
        for t in range(6):
            b = GoBoard(9, 7.0, True)
            for k in range(t):
                b.put_stone(b.onboard_pos[10 * k + t], 1 + k % 2)
            eng.set_root(t, b, 1 + t % 2, np.random.RandomState(40 + t).get_state())
        eng.root_eval(False)
        for i in range(6):
            if grow_to and i == 3:
                if fail:
                    rc = eng.lib.tg_search_grow(eng.handle, 1 << 30)          # ~ 6 x 2^30 x 3 KB: cannot be allocated
                    assert rc != 0 and b"tg_search_grow" in eng.lib.tg_last_error()
                tl.check(eng.lib.tg_search_grow(eng.handle, grow_to), "tg_search_grow")
                eng.N = grow_to
            eng.puct_batch(8)
        h = hashlib.sha256()
        st = eng.read_root_stats()
        for k in sorted(st):
            h.update(np.ascontiguousarray(st[k]).tobytes())
        nn = eng.num_nodes()
        h.update(nn.tobytes())
        for t in range(6):
            for node in range(0, int(nn[t]), 5):
                nd = eng.read_node(t, node)
                n = nd.num_children
                for arr in (nd.children_index[:n], nd.children_visits[:n], nd.children_value_sum[:n], nd.children_policy[:n]):
                    h.update(np.ascontiguousarray(arr).tobytes())
        eng.close()
        return h.hexdigest(), int(nn.sum())

    want = run(128)
    assert want[1] > 6 * 40
    assert run(32, grow_to=128) == want                       # (24 nodes per tree in use when the pool grows)
    assert run(32, grow_to=128, fail=True) == want


def test_batch_queue_is_the_device_leaf_queue():
    """mcts/batch_data.py:7-34: between selection and process_mini_batch the queue holds the
    leaves' planes, root-first paths and node indices - the same entries the oracle's queue holds -
    and it is empty after the flush and after every search call."""
    from oracle.board import GoBoard as OBoard
    from oracle.stubnet import StubNet
    from oracle.tree import MCTSTree as OTree
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.mcts.batch_data import BatchQueue
    from tamago_amd.mcts.engine import SearchEngine, HostEvaluator
    from tamago_amd.mcts.time_manager import TimeManager, TimeControl
    from tamago_amd.mcts.tree import MCTSTree
    q = BatchQueue()                                           # the container itself
    q.push(np.zeros((6, 9, 9), np.float32), [(0, 3)], 7)
    assert (len(q), q.path, q.node_index) == (1, [[(0, 3)]], [7]) and q.input_plane[0].shape == (6, 9, 9)
    q.clear()
    assert len(q) == 0 and q.node_index == []

    eng = SearchEngine(9, 1, 128, 8, HostEvaluator(StubNet(6), torch.device("cuda:0")))
    eng.set_root(0, GoBoard(9), 1, np.random.RandomState(21).get_state())
    eng.root_eval(False)
    oracle = OTree(StubNet(6), 9, tree_size=128, batch_size=64)     # large batch: no flush inside
    np.random.seed(21)
    board = OBoard(9)
    oracle._initialize_search(board, 1)
    for rounds in range(3):
        eng.puct_select(8)
        got = eng.read_queue(0)
        scratch = board.clone()
        for _ in range(8):
            scratch.copy_from(board)
            oracle.search_mcts(scratch, 1, oracle.current_root, [])
        want = oracle.batch_queue
        assert len(got) == len(want.node_index) == 8
        # a leaf that was already expanded goes to the reference's node[-1]: -1 here, tree_size-1 there
        assert [i if i >= 0 else 127 for i in got.node_index] == \
            [i % 128 for i in want.node_index]
        assert got.path == [[(int(a), int(b)) for a, b in p] for p in want.path]
        assert np.array_equal(np.array(got.input_plane), np.array(want.input_plane, dtype=np.float32))
        eng.puct_flush()
        oracle.process_mini_batch(board)
        assert len(eng.read_queue(0)) == 0
    eng.close()

    tree = MCTSTree(StubNet(6), tree_size=128, batch_size=8)
    assert len(tree.batch_queue) == 0
    np.random.seed(2)
    tree.search_best_move(GoBoard(9), 1, TimeManager(TimeControl.STRICT_PLAYOUT, 30), {})
    assert len(tree.batch_queue) == 0 and tree.batch_queue.node_index == []


def test_load_network_good_file_missing_file_and_device(tmp_path, capsys):
    """nn/utility.py:139-159: a readable state_dict is loaded, an unreadable path keeps the random
    initialisation and prints the reference's message, the network lives on the requested device."""
    from oracle.net import OracleNet, make_state_dict
    from tamago_amd.nn.utility import load_network
    sd = make_state_dict(9, 5, 1.2)
    path = tmp_path / "model.bin"
    torch.save(sd, path)
    x = torch.from_numpy(np.random.RandomState(0).randint(-1, 2, size=(4, 6, 9, 9)).astype(np.float32))
    net = load_network(model_file_path=str(path), use_gpu=True, board_size=9, device_index=0)
    assert capsys.readouterr().out == ""
    assert net.device == torch.device("cuda", 0) and net.device_index == 0
    pol, val = net.inference(x)
    rpol, rval = OracleNet(sd).inference(x)
    assert float((pol - rpol).abs().max()) < 1e-4 and float((val - rval).abs().max()) < 1e-4
    torch.manual_seed(77)
    rnd = load_network(model_file_path=str(tmp_path / "missing.bin"), use_gpu=True, board_size=9)
    assert capsys.readouterr().out == f"Failed to load {tmp_path / 'missing.bin'}.\n"
    pol2, _ = rnd.inference(x)
    assert torch.isfinite(pol2).all() and float((pol2 - pol).abs().max()) > 1e-6       # not the file's weights
    torch.manual_seed(77)
    from tamago_amd.nn.network.dual_net import random_state_dict
    want = OracleNet(random_state_dict(9)).inference(x)[0]                                # the kept random init
    assert float((pol2 - want).abs().max()) < 1e-4
    (tmp_path / "garbage.bin").write_bytes(b"not a checkpoint")
    load_network(model_file_path=str(tmp_path / "garbage.bin"), use_gpu=True, board_size=9)
    assert "Failed to load" in capsys.readouterr().out
    with pytest.raises(RuntimeError):
        load_network(model_file_path=str(path), use_gpu=False)
    if torch.cuda.device_count() > 1:
        assert load_network(str(path), True, 9, device_index=1).device_index == 1


def test_board_size_mismatch_is_rejected():
    """A 9x9 network cannot serve a 19x19 tree: the reference's layers raise a shape error; here
    the device kernels would read a [B,82] policy with a [B,362] stride."""
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.gtp.client import GtpClient
    from tamago_amd.mcts.time_manager import TimeManager, TimeControl
    from tamago_amd.mcts.tree import MCTSTree
    from tamago_amd.nn.network.dual_net import DualNet
    net = DualNet(torch.device("cuda:0"), 9)
    with pytest.raises(ValueError):
        net.forward_device(torch.zeros((2, 6, 19, 19), device="cuda:0"))
    with pytest.raises(ValueError):
        net.inference(torch.zeros((2, 6, 19, 19)))
    tree = MCTSTree(net, tree_size=64, batch_size=4)
    with pytest.raises(ValueError):
        tree.search_best_move(GoBoard(19), 1, TimeManager(TimeControl.STRICT_PLAYOUT, 8), {})
    out = io.StringIO()
    client = GtpClient(9, False, net, stdout=out)
    client.command_id = ""
    client.commands["boardsize"](["19"])
    assert out.getvalue().startswith("?")
    out.truncate(0), out.seek(0)
    client.commands["boardsize"](["9"])
    assert out.getvalue().startswith("=")


def test_19x19_forwards_on_two_streams_do_not_share_scratch():
    """The 19x19 Winograd kernel passes its layer activations through a scratch image in global
    memory; forwards of one network launched from two threads on two streams (self-play groups)
    must each get their own.  Every result must equal the single-stream result bit for bit."""
    from tamago_amd.nn.network.dual_net import DualNet
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    net = DualNet(dev, 19)
    rs = np.random.RandomState(3)
    xs = [torch.from_numpy(rs.randint(-1, 2, size=(b, 6, 19, 19)).astype(np.float32)).to(dev)
          for b in (64, 48)]
    want = [tuple(t.clone() for t in net.forward_device(x, True)) for x in xs]
    torch.cuda.synchronize()
    bad = []

    def work(i):
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            for _ in range(60):
                pol, val = net.forward_device(xs[i], True)
                stream.synchronize()
                if not (torch.equal(pol, want[i][0]) and torch.equal(val, want[i][1])):
                    bad.append(i)
                    return

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not bad


def test_19x19_selfplay_groups_equal_single_group(tmp_path):
    """19x19 shard with two pipelined groups sharing one DualNet = the same games as one group."""
    from tamago_amd.nn.network.dual_net import DualNet
    from tamago_amd.selfplay.worker import selfplay_shard
    torch.manual_seed(6)
    net = DualNet(torch.device("cuda:0"), 19)
    idx = [1, 2, 3, 4]
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir(), two.mkdir()

    def short(d, groups):
        import tamago_amd.selfplay.worker as w
        return selfplay_shard(str(d), net, idx, 19, 16, boards=4, never_resign_flags=[False] * 4, groups=groups)

    a = short(one, 1)
    b = short(two, 2)
    assert a["games"] == b["games"] == 4
    for i in idx:
        assert open(one / f"{i}.sgf").read() == open(two / f"{i}.sgf").read(), i
