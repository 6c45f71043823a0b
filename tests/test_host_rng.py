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
