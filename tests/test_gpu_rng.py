"""The device-resident legacy random streams (csrc/legacy_rng_device.h: MT19937, random_sample, glibc's log restated;
mcts/tree.py:509-519 np.random.dirichlet, mcts/node.py:275-278 np.random.gumbel) against numpy's RandomState on the box and
against the reference-recorded draws: windows whole and in pieces, every workgroup shape of the generation kernel, start
positions inside / at the end of a state block and odd ones (a double straddling two blocks), consumption committed between
windows, the state handed back to numpy."""
import ctypes

import numpy as np
import pytest

from tests.helpers import load_npz

pytestmark = pytest.mark.gpu


def _engine(trees, size=9):
    import torch
    from tamago_amd.mcts.engine import SearchEngine, DeviceEvaluator
    from tamago_amd.nn.network.dual_net import DualNet
    from tamago_amd.board.go_board import GoBoard
    net = DualNet(torch.device("cuda:0"), size)
    eng = SearchEngine(size, trees, 64, 8, DeviceEvaluator(net))
    board = GoBoard(size, 7.0, False)
    return eng, board


def _states(trees):
    """numpy states with every kind of start position: fresh seed (pos 624), mid-block even, odd (after randint draws of one
    word each), one word before the end of the block."""
    out = []
    for t in range(trees):
        rs = np.random.RandomState(1000 + t)
        kind = t % 4
        if kind == 1:
            rs.random_sample(7 + t)
        elif kind == 2:
            rs.randint(0, 2 ** 31 - 1, size=2 * t + 1)           # an odd number of 32-bit words
        elif kind == 3:
            rs.random_sample(311)
            rs.randint(0, 2 ** 31 - 1, size=1)                   # pos = 623: the next double straddles the regeneration
        out.append(rs.get_state())
    return out


def _read_window(eng, tree, first, count):
    from tamago_amd import lib as tl
    out = np.empty(count, dtype=np.float64)
    tl.check(eng.lib.tg_search_debug_read_window(eng.handle, tree, first, count, out.ctypes.data), "tg_search_debug_read_window")
    return out


def _walk(eng, steps, slack, part):
    from tamago_amd import lib as tl
    arr = np.ascontiguousarray(steps, dtype=np.int64)
    tl.check(eng.lib.tg_search_debug_stream_walk(eng.handle, arr.ctypes.data, len(arr), slack, part), "tg_search_debug_stream_walk")


@pytest.mark.parametrize("trees", [1, 5, 70, 520])                   # 16 / 16 / 4 / 1 wavefronts per tree
def test_windows_equal_numpy_standard_exponential(trees):
    eng, board = _engine(trees)
    try:
        states = _states(trees)
        for t, st in enumerate(states):
            eng.set_root(t, board, 1, st)
        ref = []
        for st in states:
            g = np.random.RandomState()
            g.set_state(st)
            ref.append(g)
        sample = sorted(set([0, 1, 2, 3, trees // 2, trees - 1]) & set(range(trees)))
        # whole windows of several sizes (a fraction of a block ... many blocks), 40 % of each consumed
        for need in (5, 311, 312, 313, 1000, 21000 if trees <= 70 else 3000):
            used = (need * 2) // 5
            _walk(eng, [used], need - used, 0)
            for t in sample:
                g = np.random.RandomState()
                g.set_state(ref[t].get_state())
                assert np.array_equal(_read_window(eng, t, 0, need), g.standard_exponential(need)), (trees, need, t)
            for g in ref:
                g.standard_exponential(used)
        # a window in pieces: first part, continued pieces, rest - the same draws as one piece
        need, part = (50000, 6000) if trees <= 70 else (5000, 700)
        _walk(eng, [need // 3], need - need // 3, part)
        for t in sample:
            g = np.random.RandomState()
            g.set_state(ref[t].get_state())
            assert np.array_equal(_read_window(eng, t, 0, need), g.standard_exponential(need)), (trees, "pieces", t)
        for g in ref:
            g.standard_exponential(need // 3)
        # the state at the logical position goes back to numpy
        for t in sample:
            got = np.random.RandomState()
            got.set_state(eng.streams[t].final_state())
            chk = np.random.RandomState()
            chk.set_state(ref[t].get_state())
            assert np.array_equal(got.random_sample(5), chk.random_sample(5)), (trees, t)
            a, b = eng.streams[t].final_state(), ref[t].get_state()
            assert a[2] == b[2] and np.array_equal(a[1], b[1]), (trees, t)        # incl. numpy's lazy regeneration (pos may be 624)
    finally:
        eng.close()


def test_gumbel_noise_and_reference_recorded_draws():
    """tools/gen_golden.py's draw sequence per seed - dirichlet(ones(n)) for n = 1, 2, 37, 82, 362, gumbel(82), dirichlet(5),
    random_sample(4) - on device streams: windows are read back and normalised with numpy's sequential sum, the Gumbel noise is
    tg_search_draw_noise's, the final state continues the recorded uniform draws."""
    fix = load_npz("rng.npz")
    seeds = (0, 1, 12345)
    eng, board = _engine(len(seeds))
    try:
        for t, seed in enumerate(seeds):
            eng.set_root(t, board, 1, np.random.RandomState(seed).get_state())

        def dirichlet(n):
            _walk(eng, [n], 50, 0)                                    # windows may be larger than what is consumed
            for t, seed in enumerate(seeds):
                e = _read_window(eng, t, 0, n)
                acc = 0.0
                for v in e:
                    acc += v
                assert np.array_equal(e * (1.0 / acc), fix[f"seed{seed}_dir{n}"]), (seed, n)
        for n in (1, 2, 37, 82, 362):
            dirichlet(n)
        noise = eng.set_gumbel_noise()
        for t, seed in enumerate(seeds):
            assert np.array_equal(noise[t], fix[f"seed{seed}_gum82"]), seed
        dirichlet(5)
        for t, seed in enumerate(seeds):
            g = np.random.RandomState()
            g.set_state(eng.streams[t].final_state())
            assert np.array_equal(g.random_sample(4), fix[f"seed{seed}_uni"]), seed
    finally:
        eng.close()


def test_stream_state_after_long_consumption():
    """After a search the library hands numpy's generator back at the position the search left it (tg_search_stream_state ->
    np.random.set_state; mcts/tree.py draws from the process-global generator): whatever the pattern of windows and consumption -
    fewer draws than a state block, many blocks, windows far larger than what is consumed - the state must be numpy's after the
    same number of standard_exponential draws, and the next window must continue the stream."""
    patterns = [([5], 100), ([2047, 1, 1], 0), ([2048], 0), ([2049], 7000), ([82] * 60, 21000), ([70000], 1000),
                ([1000, 0, 50000, 3, 2048 * 7], 5000), ([600000], 23000)]
    eng, board = _engine(2)
    try:
        for steps, slack in patterns:
            refs = []
            for t, seed in enumerate((3, 2 ** 31 - 9)):
                ref = np.random.RandomState(seed)
                ref.random_sample(11)                                 # (a mid-state start position)
                eng.set_root(t, board, 1, ref.get_state())
                refs.append(ref)
            _walk(eng, steps, slack, 0)
            for t, ref in enumerate(refs):
                ref.standard_exponential(int(np.sum(steps)))
                a, b = eng.streams[t].final_state(), ref.get_state()
                assert a[2] == b[2] and np.array_equal(a[1], b[1]), (steps, t)
            _walk(eng, [0], 9, 0)
            for t, ref in enumerate(refs):
                assert np.array_equal(_read_window(eng, t, 0, 9), ref.standard_exponential(9)), (steps, t)
    finally:
        eng.close()
