"""ExpStream (host side of the device RNG feed) vs numpy's legacy generator and the
reference-recorded draws - CPU only."""
import numpy as np

from tests.helpers import load_npz


def test_expstream_reproduces_dirichlet_and_gumbel_draws():
    from tamago_amd.mcts.engine import ExpStream
    fix = load_npz("rng.npz")
    for seed in (0, 1, 12345):
        s = ExpStream(np.random.RandomState(seed).get_state())
        for n in (1, 2, 37, 82, 362):
            e = s.window(n + 50)[:n].copy()          # windows may be larger than what is consumed
            s.consume(n)
            acc = 0.0
            for v in e:
                acc += v                              # sequential sum, like numpy's dirichlet
            assert np.array_equal(e * (1.0 / acc), fix[f"seed{seed}_dir{n}"]), (seed, n)
        assert np.array_equal(s.gumbel(82), fix[f"seed{seed}_gum82"])
        e = s.window(5).copy()
        s.consume(5)
        acc = 0.0
        for v in e:
            acc += v
        assert np.array_equal(e * (1.0 / acc), fix[f"seed{seed}_dir5"])
        g = np.random.RandomState()
        g.set_state(s.final_state())
        assert np.array_equal(g.random_sample(4), fix[f"seed{seed}_uni"])


def test_library_legacy_stream_equals_numpy():
    """tg_legacy_exponentials (the arithmetic of the library-owned streams: MT19937, legacy
    random_sample, standard_exponential) vs numpy.random.RandomState, across state refills and
    from mid-state positions; the updated state continues numpy's stream."""
    from tamago_amd import lib as tl
    lib = tl.load()
    fix = load_npz("rng.npz")
    for seed in (0, 1, 12345, 2**31 - 5):
        ref = np.random.RandomState(seed)
        state = ref.get_state()
        key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
        pos = tl.ctypes.c_int(int(state[2]))
        for n in (1, 5, 311, 313, 1000, 2):              # 311 + 313 doubles = 2 x 624 words: refill boundary
            out = np.empty(n, dtype=np.float64)
            tl.check(lib.tg_legacy_exponentials(key.ctypes.data, tl.ctypes.byref(pos), n, out.ctypes.data))
            assert np.array_equal(out, ref.standard_exponential(n)), (seed, n)
        cont = np.random.RandomState()
        cont.set_state(("MT19937", key, int(pos.value), 0, 0.0))
        assert np.array_equal(cont.random_sample(7), ref.random_sample(7))
    # the recorded reference draws: dirichlet(ones(n)) = normalised exponentials, gumbel = -log(e)
    for seed in (0, 1, 12345):
        state = np.random.RandomState(seed).get_state()
        key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
        pos = tl.ctypes.c_int(int(state[2]))
        for n in (1, 2, 37, 82, 362):
            e = np.empty(n)
            tl.check(lib.tg_legacy_exponentials(key.ctypes.data, tl.ctypes.byref(pos), n, e.ctypes.data))
            acc = 0.0
            for v in e:
                acc += v
            assert np.array_equal(e * (1.0 / acc), fix[f"seed{seed}_dir{n}"]), (seed, n)
        e = np.empty(82)
        tl.check(lib.tg_legacy_exponentials(key.ctypes.data, tl.ctypes.byref(pos), 82, e.ctypes.data))
        import math
        assert np.array_equal(np.array([-math.log(v) for v in e]), fix[f"seed{seed}_gum82"])   # libm log, not numpy's SIMD log


def test_stream_state_after_long_consumption_comes_from_snapshots():
    """After a search the library hands numpy's generator back at the position the search left it (tg_search_stream_state ->
    np.random.set_state; mcts/tree.py draws from the process-global generator).  The state is rebuilt from snapshots taken every
    2 048 generated draws (csrc/legacy_stream.h) instead of replaying every consumed draw: whatever the pattern of windows and
    consumption - fewer draws than a snapshot interval, many intervals, windows far larger than what is consumed - the state
    must be numpy's after the same number of standard_exponential draws, and the staged draws must continue the stream."""
    from tamago_amd import lib as tl
    lib = tl.load()
    patterns = [([5], 100), ([2047, 1, 1], 0), ([2048], 0), ([2049], 7000), ([82] * 300, 21000), ([70000], 1000),
                ([1000, 0, 50000, 3, 2048 * 7], 5000), ([600000], 23000)]
    for seed in (3, 2**31 - 9):
        for steps, slack in patterns:
            ref = np.random.RandomState(seed)
            ref.random_sample(11)                                 # (a mid-state start position)
            state = ref.get_state()
            key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
            arr = np.ascontiguousarray(steps, dtype=np.int64)
            key_out = np.zeros(624, dtype=np.uint32)
            pos_out = tl.ctypes.c_int(0)
            nxt = np.empty(9, dtype=np.float64)
            tl.check(lib.tg_legacy_stream_walk(key.ctypes.data, int(state[2]), arr.ctypes.data, len(arr), slack,
                                               key_out.ctypes.data, tl.ctypes.byref(pos_out), nxt.ctypes.data, len(nxt)),
                     "tg_legacy_stream_walk")
            ref.standard_exponential(int(arr.sum()))
            want = ref.get_state()
            got = np.random.RandomState()
            got.set_state(("MT19937", key_out, int(pos_out.value), 0, 0.0))
            chk = np.random.RandomState()
            chk.set_state(want)
            assert np.array_equal(got.random_sample(6), chk.random_sample(6)), (seed, steps)
            chk.set_state(want)
            assert np.array_equal(nxt, chk.standard_exponential(9)), (seed, steps)
