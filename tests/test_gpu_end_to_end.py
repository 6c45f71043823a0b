"""End-to-end hot path on the GPU at BASELINE.json's full size (9x9, 1000 strict visits,
NN batch 256; and 19x19, batch 64): PUCT selection, leaf featurisation, the fused HIP
DualNet forward, expansion and backup all run on the device.

A GPU forward pass is only within 1e-4 of the CPU one, and a 1e-7 difference already moves
visit counts, so exactness is shown like this: every mini-batch the GPU
evaluated is recorded (planes + outputs) and replayed, in order, into the CPU oracle tree.
The oracle must (1) ask for bit-identical leaf planes in the same order - i.e. selection,
board engine, expansion and featurisation agree leaf by leaf -, (2) see its own CPU network
agree with the recorded GPU outputs within 1e-4, and (3) end with identical visit counts,
value sums and node count."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class Recorder:
    def __init__(self, inner):
        self.inner = inner
        self.log = []

    def __call__(self, planes, want_logits):
        policy, value = self.inner(planes, want_logits)
        self.log.append((planes.cpu().clone(), policy.cpu().clone(), value.cpu().clone(), want_logits))
        return policy, value


class ReplayNet:
    """Oracle-side network: returns the recorded GPU outputs, checks planes and CPU parity."""

    def __init__(self, log, cpu_net):
        self.log = list(log)
        self.cpu_net = cpu_net
        self.i = 0
        self.max_err = 0.0

    def _next(self, planes, want_logits):
        rec_planes, policy, value, logits = self.log[self.i]
        self.i += 1
        assert logits == want_logits
        assert torch.equal(planes, rec_planes), f"leaf planes differ in mini-batch {self.i - 1}"
        ref_p, ref_v = (self.cpu_net.inference_with_policy_logits(planes) if want_logits
                        else self.cpu_net.inference(planes))
        if not want_logits:
            self.max_err = max(self.max_err, float((ref_p - policy).abs().max()))
        self.max_err = max(self.max_err, float((ref_v - value).abs().max()))
        return policy, value

    def inference(self, planes):
        return self._next(planes, False)

    def inference_with_policy_logits(self, planes):
        return self._next(planes, True)


def _run(size, visits, batch, plies, seed, gumbel=False):
    from oracle.board import GoBoard as OBoard
    from oracle.net import OracleNet, make_state_dict
    from oracle.tree import MCTSTree as OTree, TimeManager as OTM, TimeControl as OTC
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.mcts import tree as ptree
    from tamago_amd.mcts.time_manager import TimeManager, TimeControl
    from tamago_amd.nn.network.dual_net import DualNet
    from tests.helpers import load_npz

    sd = make_state_dict(size, 7, 1.5)
    net = DualNet(torch.device("cuda:0"), size)
    net.load_state_dict(sd)
    brd = load_npz(f"board_s{size}.npz")
    board, oboard = GoBoard(size, 7.0, True), OBoard(size, 7.0, True)
    for mv, c in zip(brd["g0_move"][:plies], brd["g0_color"][:plies]):
        board.put_stone(int(mv), int(c))
        oboard.put_stone(int(mv), int(c))
    color = 1 if plies == 0 else 3 - int(brd["g0_color"][plies - 1])

    class RecordingTree(ptree.MCTSTree):
        def _evaluator(self):
            self.recorder = Recorder(super()._evaluator())
            return self.recorder

    tree = RecordingTree(net, tree_size=visits + 64, batch_size=batch)
    np.random.seed(seed)
    if gumbel:
        mv = tree.generate_move_with_sequential_halving(
            board, color, TimeManager(TimeControl.CONSTANT_PLAYOUT, visits), True)
    else:
        mv = tree.search_best_move(board, color, TimeManager(TimeControl.STRICT_PLAYOUT, visits), {})
    rng_after = float(np.random.random_sample())

    replay = ReplayNet(tree.recorder.log, OracleNet(sd))
    otree = OTree(replay, size, tree_size=visits + 64, batch_size=batch)
    np.random.seed(seed)
    if gumbel:
        omv = otree.generate_move_with_sequential_halving(
            oboard, color, OTM(OTC.CONSTANT_PLAYOUT, visits), True)
    else:
        omv = otree.search_best_move(oboard, color, OTM(OTC.STRICT_PLAYOUT, visits))
    assert replay.i == len(replay.log)
    assert float(np.random.random_sample()) == rng_after
    root, oroot = tree.get_root(), otree.get_root()
    n = oroot.num_children
    assert mv == omv and root.num_children == n and tree.num_nodes == otree.num_nodes
    assert np.array_equal(root.children_visits[:n], oroot.children_visits[:n])
    assert np.array_equal(root.children_value_sum[:n], oroot.children_value_sum[:n])
    assert np.array_equal(root.children_policy[:n], oroot.children_policy[:n])
    assert replay.max_err < 1e-4, replay.max_err
    return tree, root, replay


def test_cfg2_puct_9x9_batch256_1000_visits():
    tree, root, replay = _run(9, 1000, 256, 24, seed=3)
    assert [b[0].shape[0] for b in tree.recorder.log] == [1, 256, 256, 256, 232]
    # size-independent invariants of a finished strict search
    assert root.node_visits == 1000 and int(root.children_visits.sum()) == 1000
    assert 900 < tree.num_nodes <= 1001 and root.virtual_loss == 0   # re-queued leaves add no node
    assert int(np.abs(root.children_virtual_loss).sum()) == 0


def test_cfg5_puct_19x19_batch64_1600_visits():
    """BASELINE.json config 5 at its full size: 19x19, 1600 strict visits, NN batch 64."""
    tree, root, _ = _run(19, 1600, 64, 60, seed=4)
    assert [b[0].shape[0] for b in tree.recorder.log] == [1] + [64] * 25
    assert root.node_visits == 1600 and 1400 < tree.num_nodes <= 1601
    assert int(root.children_visits.sum()) == 1600 and root.virtual_loss == 0


def test_multi_tree_lockstep_device_evaluator_replay():
    """64 trees in lock-step on the DeviceEvaluator path (what bench.py and the self-play shards
    run): every tree's slice of every recorded mini-batch is replayed into its own CPU oracle
    tree - identical leaf planes in order, identical visit counts / value sums / node counts."""
    from oracle.board import GoBoard as OBoard
    from oracle.net import OracleNet, make_state_dict
    from oracle.tree import MCTSTree as OTree, TimeManager as OTM, TimeControl as OTC
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.mcts.engine import SearchEngine, DeviceEvaluator
    from tamago_amd.nn.network.dual_net import DualNet
    from tests.helpers import load_npz

    size, T, K, visits = 9, 64, 64, 200
    sd = make_state_dict(size, 7, 1.5)
    net = DualNet(torch.device("cuda:0"), size)
    net.load_state_dict(sd)
    rec = Recorder(DeviceEvaluator(net))
    engine = SearchEngine(size, T, visits + 16, K, rec)
    brd = load_npz("board_s9.npz")
    roots = []
    for t in range(T):
        game, plies = t % 4, 2 + (t * 5) % 40
        board, oboard = GoBoard(size, 7.0, False), OBoard(size, 7.0, False)
        mv, col = brd[f"g{game}_move"], brd[f"g{game}_color"]
        plies = min(plies, len(mv) - 1)
        for m, c in zip(mv[:plies], col[:plies]):
            board.put_stone(int(m), int(c))
            oboard.put_stone(int(m), int(c))
        color = 3 - int(col[plies - 1])
        engine.set_root(t, board, color, np.random.RandomState(900 + t).get_state())
        roots.append((oboard, color))
    engine.root_eval(False)
    done = 0
    while done < visits:
        k = min(K, visits - done)
        engine.puct_batch(k)
        done += k
    stats = engine.read_root_stats()
    nodes = engine.num_nodes()
    assert [b[0].shape[0] for b in rec.log] == [T, T * 64, T * 64, T * 64, T * 8]
    # CPU parity of the recorded device outputs, whole mini-batches at once
    cpu = OracleNet(sd)
    for planes, policy, value, _ in rec.log:
        ref_p, ref_v = cpu.inference(planes)
        assert float((ref_p - policy).abs().max()) < 1e-4 and float((ref_v - value).abs().max()) < 1e-4

    class Slice:
        """the recorded outputs of tree t, mini-batch by mini-batch"""
        def __init__(self, t):
            self.t, self.i = t, 0

        def inference(self, planes):
            rp, pol, val, _ = rec.log[self.i]
            per = rp.shape[0] // T
            self.i += 1
            lo = self.t * per
            assert torch.equal(planes, rp[lo:lo + planes.shape[0]]), (self.t, self.i - 1)
            return pol[lo:lo + planes.shape[0]], val[lo:lo + planes.shape[0]]

    for t in range(0, T, 3):                       # every third tree: 22 oracle searches
        oboard, color = roots[t]
        sl = Slice(t)
        otree = OTree(sl, size, tree_size=visits + 16, batch_size=K)
        np.random.set_state(np.random.RandomState(900 + t).get_state())
        otree.search_best_move(oboard, color, OTM(OTC.STRICT_PLAYOUT, visits))
        oroot = otree.get_root()
        n = oroot.num_children
        assert sl.i == len(rec.log) and int(stats["num_children"][t]) == n and int(nodes[t]) == otree.num_nodes
        assert np.array_equal(stats["children_visits"][t][:n], oroot.children_visits[:n]), t
        assert np.array_equal(stats["children_value_sum"][t][:n], oroot.children_value_sum[:n]), t
        assert np.array_equal(stats["children_policy"][t][:n], oroot.children_policy[:n]), t
    engine.close()


def test_gumbel_400_sims_with_real_network():
    tree, root, _ = _run(9, 400, 1, 10, seed=5, gumbel=True)
    assert [b[0].shape[0] for b in tree.recorder.log] == [1, 96, 96, 100, 108]
    assert root.node_visits == 400
