"""Search parameters with the reference's names and values (mcts/constant.py:5-41), grouped by
the part of the search that reads them."""

_NODE_POOL = dict(
    MCTS_TREE_SIZE=1 << 16,          # nodes per tree unless the caller says otherwise
    NOT_EXPANDED=-1,                 # children_index of an edge without a node
)
_PUCT = dict(
    PUCB_SECOND_TERM_WEIGHT=1.0,
    NN_BATCH_SIZE=1,                 # leaves per mini-batch unless the caller says otherwise
    RESIGN_THRESHOLD=0.05,           # resign below this root win rate
)
_GUMBEL = dict(
    C_VISIT=50,
    C_SCALE=1.0,
    MAX_CONSIDERED_NODES=16,
    PLAYOUTS=100,                    # count threshold of the final root selection
)
_TIME = dict(
    CONST_VISITS=1000,
    CONST_TIME=5.0,
    REMAINING_TIME=60.0,
    VISITS_PER_SEC=20,               # speed estimate before the first search
)

for _group in (_NODE_POOL, _PUCT, _GUMBEL, _TIME):
    globals().update(_group)
__all__ = [name for _group in (_NODE_POOL, _PUCT, _GUMBEL, _TIME) for name in _group]
