"""One CPU self-play worker of the oracle (TEST INFRASTRUCTURE - see oracle/__init__.py): the restatement of
selfplay/worker.py:21-90's loop (Gumbel sequential-halving search per move, games to completion) on the CPU oracle, timed for
a bounded number of seconds.  bench.py's `cpu_selfplay` leg starts N of these as separate OS processes - the reference's own
parallelism (selfplay_main.py:58-65: one selfplay_worker process per --process, default NUM_SELF_PLAY_WORKERS = 4,
learning_param.py:43) - and adds up their leaf evaluations.

    python -m oracle.cpu_selfplay --seconds 8 --visits 400 --seed 1 --threads 2   ->  one JSON line
"""
import argparse
import json
import sys
import time

import numpy as np
import torch


def run(seconds: float, visits: int, seed: int, threads: int, size: int = 9) -> dict:
    from oracle.board import GoBoard, BLACK
    from oracle.net import OracleNet, make_state_dict
    from oracle.tree import MCTSTree, TimeManager, TimeControl
    torch.set_num_threads(max(1, threads))
    net = OracleNet(make_state_dict(size, 0, 1.0))
    tree = MCTSTree(net, size, tree_size=visits * 10, batch_size=10 ** 9)     # worker.py:41 (batches are the halving phases')
    tm = TimeManager(TimeControl.CONSTANT_PLAYOUT, visits)
    np.random.seed(seed)                                                      # worker.py:38
    leaves = moves = games = 0
    t0 = time.time()
    deadline = t0 + seconds
    while time.time() < deadline:
        board = GoBoard(size, 7.0, True)
        color, passes = BLACK, 0
        for _ in range(2 * size * size):
            if time.time() >= deadline:
                break
            pos = tree.generate_move_with_sequential_halving(board, color, tm, True)
            leaves += sum(tree.batch_log)
            tree.batch_log.clear()
            board.put_stone(pos if pos > 0 else 0, color)
            moves += 1
            passes = passes + 1 if pos == 0 else 0
            color = 3 - color
            if passes == 2:
                games += 1
                break
    dt = time.time() - t0
    return {"leaf_evals": leaves, "seconds": dt, "moves": moves, "games_finished": games, "threads": torch.get_num_threads()}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--visits", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--threads", type=int, default=1)
    a = ap.parse_args()
    print(json.dumps(run(a.seconds, a.visits, a.seed, a.threads)))
    sys.stdout.flush()
