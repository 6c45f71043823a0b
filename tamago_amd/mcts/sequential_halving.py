"""Sequential-halving schedule (host integer logic; mirror of the mctx schedule used by
mcts/sequential_halving.py:7-60)."""
import functools
import math
from typing import Dict, Tuple


def get_sequence_of_considered_visits(max_num_considered_actions: int,
                                      num_simulations: int) -> Tuple[int, ...]:
    if max_num_considered_actions <= 1:
        return tuple(range(num_simulations))
    log2max = int(math.ceil(math.log2(max_num_considered_actions)))
    out = []
    visits = [0] * max_num_considered_actions
    width = max_num_considered_actions
    while len(out) < num_simulations:
        rounds = max(1, int(num_simulations / (log2max * width)))
        for _ in range(rounds):
            out.extend(visits[:width])
            visits[:width] = [v + 1 for v in visits[:width]]
        width = max(2, width // 2)
    return tuple(out[:num_simulations])


def get_candidates_and_visit_pairs(max_num_considered_actions: int,
                                   num_simulations: int) -> Dict[int, int]:
    """{number of considered actions: number of levels} in phase order."""
    return dict(_pairs_cached(max_num_considered_actions, num_simulations))


@functools.lru_cache(maxsize=256)
def _pairs_cached(max_num_considered_actions: int, num_simulations: int):
    seq = get_sequence_of_considered_visits(max_num_considered_actions, num_simulations)
    width_at_level = [0] * (max(seq) + 1)
    for level in seq:
        width_at_level[level] += 1
    pairs: Dict[int, int] = {}
    for width in width_at_level:
        pairs[width] = pairs.get(width, 0) + 1
    return tuple(pairs.items())
