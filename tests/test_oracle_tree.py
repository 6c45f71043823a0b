"""Oracle search (PUCT + Gumbel sequential halving) vs reference-generated fixtures.

The evaluator is oracle.stubnet.StubNet on both sides, so every number must match
bit-for-bit: visit counts, float32-accumulated value sums, chosen move, node count,
batch sizes, a digest over the whole tree, and the position of the global RNG stream
after the search."""
import hashlib

import numpy as np
import pytest

from oracle.halving import candidates_and_visit_pairs
from oracle.stubnet import StubNet
from oracle.tree import MCTSTree, TimeManager, TimeControl
from tests.helpers import load_json, load_npz, oracle_replay, unhex


def tree_digest(tree):
    h = hashlib.sha256()
    for i in range(tree.num_nodes):
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


def check_root(tree, mv, rec):
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
    assert tree_digest(tree) == rec["digest"]


def cases(size, kind):
    return [r for r in load_json(f"trees_s{size}.json") if r["kind"] == kind]


def test_halving_schedule():
    for key, pairs in load_json("tables.json")["halving"].items():
        n0, v = (int(s) for s in key.split(","))
        assert [[k, c] for k, c in candidates_and_visit_pairs(n0, v).items()] == pairs


@pytest.mark.parametrize("size", [9, 13, 19])
def test_puct(size):
    brd = load_npz(f"board_s{size}.npz")
    for rec in cases(size, "puct"):
        board = oracle_replay(size, brd["g0_move"], brd["g0_color"], rec["ply"], rec["superko"])
        net = StubNet(salt=rec["seed"])
        tree = MCTSTree(net, size, tree_size=2048, batch_size=rec["batch"], cgos_mode=rec["cgos"])
        mode = TimeControl.STRICT_PLAYOUT if rec["mode"] == "STRICT" else TimeControl.CONSTANT_PLAYOUT
        np.random.seed(rec["seed"])
        mv = tree.search_best_move(board, rec["color"], TimeManager(mode, rec["visits"]))
        assert net.calls == rec["batches"]
        check_root(tree, mv, rec)
        assert float(np.random.random_sample()) == float.fromhex(rec["rng_after"])


@pytest.mark.parametrize("size", [9, 13, 19])
def test_gumbel(size):
    brd = load_npz(f"board_s{size}.npz")
    for rec in cases(size, "gumbel"):
        board = oracle_replay(size, brd["g0_move"], brd["g0_color"], rec["ply"], rec["superko"])
        net = StubNet(salt=100 + rec["seed"])
        tree = MCTSTree(net, size, tree_size=160 if rec["visits"] <= 100 else 2048)
        np.random.seed(rec["seed"])
        mv = tree.generate_move_with_sequential_halving(
            board, rec["color"], TimeManager(TimeControl.CONSTANT_PLAYOUT, rec["visits"]), True)
        assert net.calls == rec["batches"]
        check_root(tree, mv, rec)
        root = tree.get_root()
        assert np.array_equal(root.noise, unhex(rec["noise"]))
        assert np.array_equal(root.improved_policy(), unhex(rec["improved"]))
        assert float(np.random.random_sample()) == float.fromhex(rec["rng_after"])


@pytest.mark.parametrize("visits", [16, 100, 400])
def test_random_draws_of_a_gumbel_phase_stay_inside_the_provisioned_window(visits):
    """What tg_selfplay_play_move relies on when it uploads ONE random window for all phases of a move
    (csrc/search.hip): in a sequential-halving phase only the first descent through a root child can expand a
    node, and a phase enters at most `width` root children it has not entered before in this move - so the
    expansions (one Dirichlet prior of <= A draws each) of phase p are bounded by
    min(slots_p, A, width_p + entered_before).  Checked on the oracle (= the reference's algorithm) over seeded
    mid-game positions; the device consumes draws exactly as the oracle does (tree fixtures, RNG position)."""
    from oracle.board import GoBoard, BLACK, WHITE
    from oracle.halving import candidates_and_visit_pairs as pairs_of

    class Counting(MCTSTree):
        def search_by_sequential_halving(self, board, color, threshold):
            search_board = board.clone()
            n_root = self.node[self.current_root].num_children
            base = n_root if n_root < 16 else 16
            seen = 0
            A = board.board_size ** 2 + 1
            root = self.node[self.current_root]
            for num_considered, max_count in pairs_of(base, threshold).items():
                before_nodes = self.num_nodes
                visited_before = {e for e in range(root.num_children) if root.children_visits[e] + root.children_virtual_loss[e] > 0}
                for count_threshold in range(max_count):
                    for _ in range(num_considered):
                        search_board.copy_from(board)
                        self.search_sequential_halving(search_board, color, self.current_root, [], count_threshold + 1)
                entered = {e for e in range(root.num_children) if root.children_virtual_loss[e] > 0}
                expansions = self.num_nodes - before_nodes
                slots = num_considered * max_count
                bound = min(slots, A, num_considered + seen)            # the window the library provisions (x A draws)
                assert expansions <= len(entered) <= bound, (num_considered, max_count, expansions, len(entered), bound)
                assert len(entered - visited_before) <= num_considered
                seen = min(A, seen + bound)
                self.process_mini_batch(search_board, use_logit=True)

    rs = np.random.RandomState(1234 + visits)
    for game in range(4):
        board = GoBoard(9, check_superko=True)
        color = BLACK
        for _ in range(rs.randint(0, 40)):                              # a seeded random opening
            legal = [p for p in board.onboard_pos if board.is_legal(p, color)]
            if not legal:
                break
            board.put_stone(legal[rs.randint(len(legal))], color)
            color = WHITE if color == BLACK else BLACK
        np.random.seed(77 + game)
        tree = Counting(StubNet(3 + game), 9, tree_size=visits * 10 + 16, batch_size=max(visits, 1))
        tree.generate_move_with_sequential_halving(board, color, TimeManager(TimeControl.CONSTANT_PLAYOUT, visits), True)


def test_root_choices_of_a_halving_phase_as_prefix_sums():
    """What select_gumbel_pipe_kernel's selector relies on (csrc/search.hip, "the root choices of the phase"): node.py:324-346 picks,
    for each of `width` descents of a threshold level, the first child in score order whose visits + virtual losses are under
    the threshold (child 0 if none) - the scores do not move within a phase.  A level therefore hands its descents out greedily
    in rank order: only the first `width` children under the threshold can take any, child i takes min(threshold - count_i,
    what is left), the rest go to child 0; and once `width` children stand exactly at the threshold with nothing ranked before
    the last of them able to come under a later one, every remaining level repeats itself.  One choice at a time against
    that, on random counters."""
    rs = np.random.RandomState(11)

    def one_by_one(cnt, rank, width, levels):
        cnt = cnt.copy()
        picks = []
        for th in range(1, levels + 1):
            for _ in range(width):
                under = [c for c in rank if cnt[c] < th]
                c = under[0] if under else 0            # every score -10000: np.argmax returns child 0
                cnt[c] += 1
                picks.append(c)
        return picks, cnt

    def by_levels(cnt, rank, width, levels):
        cnt = cnt.copy()
        picks = []
        th = 1
        while th <= levels:
            need = {c: max(0, th - cnt[c]) for c in rank}
            under = [c for c in rank if need[c] > 0][:width]     # every child under takes at least one
            left = width
            takers = []
            for c in under:
                take = min(need[c], left)
                if take <= 0:
                    break
                picks += [c] * take
                cnt[c] += take
                left -= take
                takers.append(c)
            picks += [0] * left
            cnt[0] += left
            th += 1
            # steady state: `width` takers at the threshold, nothing ranked before the last of them can come under later
            if left == 0 and len(takers) == width and th <= levels:
                last = max(rank.index(c) for c in takers)
                if all((cnt[c] == th - 1) if c in takers else (cnt[c] >= levels) for c in rank[:last + 1]):
                    for _ in range(th, levels + 1):
                        picks += takers
                    for c in takers:
                        cnt[c] += levels - th + 1
                    break
        return picks, cnt

    for _ in range(400):
        n = int(rs.randint(1, 83))
        width = int(rs.randint(1, 17))
        levels = int(rs.randint(1, 40))
        rank = list(rs.permutation(n))
        shape = rs.randint(0, 3)
        cnt = (rs.randint(0, levels + 4, size=n) if shape == 0 else
               rs.randint(0, 3, size=n) if shape == 1 else
               np.where(rs.rand(n) < 0.2, rs.randint(0, levels + 1, size=n), levels + 5))
        a, ca = one_by_one(cnt, rank, width, levels)
        b, cb = by_levels(cnt, rank, width, levels)
        assert a == b, (n, width, levels)
        assert np.array_equal(ca, cb)
