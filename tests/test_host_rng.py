"""The legacy random stream's arithmetic on the host - ExpStream (the numpy-side feed) and the library's restatement of MT19937 +
glibc's log (what the device streams compute) - vs numpy's legacy generator, libm and the reference-recorded draws.  CPU only."""
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


def test_restated_glibc_log_equals_libm():
    """tg_glibc_log = csrc/legacy_rng_device.h's glibc_log on the host: the function the DEVICE streams compute their
    exponentials (-log(1 - u)) and Gumbel noise (-log of those) with, restated from glibc's FMA build.  Against Python's math.log
    (a plain call of libm's log; numpy's own np.log is a SIMD routine that rounds differently) bit for bit, on the argument classes
    the streams produce - 1 - u, the exponentials, the neighbourhood of 1 with its separate polynomial - and on wide-range values.
    (numpy's legacy generators call the same libm: the next test holds the whole chain against RandomState.)"""
    import math
    from tamago_amd import lib as tl
    lib = tl.load()
    rs = np.random.RandomState(77)
    u = rs.random_sample(200_000)
    one_minus = 1.0 - u
    expo = np.array([-math.log(v) for v in one_minus[:100_000]])
    near_one = 0.93 + 0.14 * rs.random_sample(100_000)
    edges = np.array([1.0, 1.0 - 2.0 ** -53, 1.0 + 2.0 ** -52, 1.0 - 2.0 ** -4, 1.0 + float.fromhex("0x1.09p-4"), float.fromhex("0x1.6p-1"), 2.0 ** -53, 36.7, 745.0])
    wide = np.ldexp(0.5 + rs.random_sample(100_000), rs.randint(-60, 60, 100_000))
    x = np.ascontiguousarray(np.concatenate([one_minus, expo[expo > 0], near_one, edges, wide]))
    out = np.empty_like(x)
    tl.check(lib.tg_glibc_log(x.ctypes.data, x.size, out.ctypes.data), "tg_glibc_log")
    want = np.array([math.log(v) for v in x])
    bad = np.nonzero(out.view(np.uint64) != want.view(np.uint64))[0]
    assert bad.size == 0, (x[bad[:4]], out[bad[:4]], want[bad[:4]])
