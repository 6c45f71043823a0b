"""CPU oracle for the TamaGo batched-MCTS leaf-evaluation path.

TEST INFRASTRUCTURE ONLY.  This package is a from-scratch CPU restatement
(Python / NumPy / PyTorch-CPU) of the reference algorithm for the hot path named
in BASELINE.json (PUCT / Gumbel selection -> featurise -> DualNet forward ->
expand / backup).  It exists to check the HIP product path, never to be shipped
or measured as the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.

Parity pinning: the reference has no tests or golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, produced in the build container by ``tools/gen_golden.py`` (which imports
``/root/reference``) and committed as data under ``tests/golden/``.  The CPU test
suite (``tests/test_oracle_*.py``) replays every fixture through this package.

Every function cites the reference file:line it restates.
"""
