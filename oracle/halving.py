"""Oracle sequential-halving schedule (TEST INFRASTRUCTURE - see oracle/__init__.py).

Restates mcts/sequential_halving.py:7-60 (the mctx considered-visits schedule).
"""
import math
from typing import Dict, List


def considered_visits_sequence(max_considered: int, num_simulations: int) -> List[int]:
    """sequential_halving.py:7-33."""
    if max_considered <= 1:
        return list(range(num_simulations))
    log2max = int(math.ceil(math.log2(max_considered)))
    seq: List[int] = []
    visits = [0] * max_considered
    considered = max_considered
    while len(seq) < num_simulations:
        extra = max(1, int(num_simulations / (log2max * considered)))
        for _ in range(extra):
            seq.extend(visits[:considered])
            for i in range(considered):
                visits[i] += 1
        considered = max(2, considered // 2)
    return seq[:num_simulations]


def candidates_and_visit_pairs(max_considered: int, num_simulations: int) -> Dict[int, int]:
    """sequential_halving.py:36-60: {number of considered actions: number of levels},
    in insertion order (which is the phase order used by the search)."""
    seq = considered_visits_sequence(max_considered, num_simulations)
    per_level = [0] * (max(seq) + 1)
    for v in seq:
        per_level[v] += 1
    out: Dict[int, int] = {}
    for count in per_level:
        out[count] = out.get(count, 0) + 1
    return out
