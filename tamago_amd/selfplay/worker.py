"""Self-play on MI355X: many boards per GPU in lock-step, one process per GPU.

``selfplay_worker`` keeps the reference signature (selfplay/worker.py:21-22) and - for
``index_list = [k]`` - reproduces the reference game for ``np.random.seed(k)`` move for move
(Gumbel search per move, resign rule, two-pass end, count_score, SGF text).
``selfplay_shard`` is the MI355X form: ``boards`` games advance together so that every
sequential-halving phase becomes ONE forward pass over all their leaves, and shards are
independent across GPUs (selfplay_main.py:44-65: no communication but files).
"""
import os
import random
from typing import List, Sequence

import numpy as np

from tamago_amd.board.constant import PASS, RESIGN
from tamago_amd.board.go_board import GoBoard
from tamago_amd.board.stone import Stone
from tamago_amd.mcts.engine import SearchEngine, HostEvaluator, DeviceEvaluator
from tamago_amd.sgf.selfplay_record import SelfPlayRecord

SELF_PLAY_VISITS = 16           # learning_param.py:40


def shard_indices(index_list: Sequence[int], world_size: int, rank: int) -> List[int]:
    """Contiguous split of the game indices over ranks (selfplay_main.py:44-57 does the same
    over worker processes); shards differ in size by at most one game."""
    n = len(index_list)
    base, extra = divmod(n, world_size)
    start = rank * base + min(rank, extra)
    return list(index_list[start:start + base + (1 if rank < extra else 0)])


class _Game:
    def __init__(self, index: int, size: int, save_dir: str, never_resign: bool):
        self.index = index
        self.size = size
        self.board = GoBoard(board_size=size, komi=7.0, check_superko=True)   # start position
        self.history = []                        # (pos, colour) - the game itself lives on the GPU
        self.record = SelfPlayRecord(save_dir, self.board.coordinate)
        self.color = Stone.BLACK
        self.pass_count = 0
        self.never_resign = never_resign
        self.moves_played = 0
        self.done = False


def _finish(game: _Game, winner, is_resign: bool, score: float):
    game.record.set_index(game.index)
    game.record.write_record(winner, game.board.get_komi(), is_resign, score)
    game.done = True


def selfplay_shard(save_dir: str, network, index_list: Sequence[int], size: int, visits: int,
                   boards: int = 16, seeds: Sequence[int] = None, device_index: int = 0,
                   never_resign_flags: Sequence[bool] = None, groups: int = 0, observer=None, lanes: int = 0) -> dict:
    """Play the games of `index_list`, `boards` at a time.  Game i draws from its own legacy
    stream seeded with seeds[i] (default: its index), so every game equals the reference
    game a single-board worker would play with that seed.

    `groups` > 1 splits the boards into that many independent lock-step groups, each with its
    own engine, HIP stream and host thread (default: 1 below 2048 boards, else 2).  Games are
    independent, so the result does not depend on the grouping.  (Inside ONE group the library
    already overlaps the boards' tree kernels with each other's forward passes - sub-groups on
    streams of its own, tg_selfplay_play_move in include/tamago_hip.h - without extra host threads.)

    `lanes` (DualNet evaluator, no observer): the boards of a group are split into that many LANES - each its own engine,
    selfplay handle, buffers and HIP stream - all driven by the group's ONE host thread through the two halves of a chained
    move (tg_selfplay_move_begin / _end): while the host waits for one lane's records and queues its next move, the other lanes'
    moves are running, so boards of different lanes are on different moves and a lane's select / backup kernels always have
    another lane's forward pass to hide under.  0 = auto (measured table in _auto_lanes), 1 = one lock-step group as before.
    Games are independent: the result does not depend on the lanes (tests/test_gpu_fastpath.py).

    `observer` (audit hook, one group only): called as observer(engine, event) from inside
    tg_selfplay_play_move for every evaluated mini-batch and every decided move
    (tg_selfplay_set_observer in include/tamago_hip.h; event = lib.SelfplayEvent)."""
    import threading
    todo = [i for i in index_list if not os.path.isfile(os.path.join(save_dir, f"{i}.sgf"))]
    seeds = dict(zip(index_list, seeds if seeds is not None else index_list))
    flags = dict(zip(index_list, never_resign_flags)) if never_resign_flags is not None else None
    stats = {"games": 0, "moves": 0, "leaf_evals": 0, "range_fallbacks": 0}
    if not todo:
        return stats
    fb0 = network.range_fallbacks() if hasattr(network, "range_fallbacks") else 0
    boards = min(boards, len(todo))
    if groups <= 0:
        # ONE group below 2048 boards.  Measured on MI355X (400 simulations, leaf-evals/s; profiles/r06_selfplay_lanes_sweep.txt,
        # tools/experiments/sp_bench_context.py).  Round 6, random streams generated on the device: two groups - two lock-step
        # halves on their own host threads, streams and moves - reach 3.7 M at 16 boards (one group: 3.45-3.57 M) and 4.5 M at 24
        # (3.6 M) in a fresh process most of the time, but the same call repeated inside one process gives 3.20 / 3.12 / 3.19 /
        # 3.81 / 3.39 / 3.52 M where one group gives 3.54-3.57 M six times out of six, the bench's own leg landed on 3.79 / 3.16 /
        # 3.16 M, and a one-group run AFTER two-group runs in the same process dropped to 2.7 M.  Pacing the groups against each
        # other did not tame it.  The mean is below one group's: not the default.  (`groups=2` remains for callers that measure it
        # on their own workload; from 32 boards on it loses outright: 32: 4.66 -> 4.39 M, 64: 5.87 -> 4.72 M; with the draws
        # generated on host threads it lost everywhere: round 4, 16 boards 2.88 M vs 1.75 M.  At 2048 boards two groups were level
        # in round 3: 4.75 vs 4.70 M.)
        groups = 1 if boards < 2048 else 2
    groups = max(1, min(groups, boards))
    if observer is not None and groups != 1:
        raise ValueError("selfplay_shard: an observer needs groups = 1")
    queue = list(todo)
    lock = threading.Lock()

    def next_game():
        """(index, never_resign) of the next unplayed game, or None."""
        with lock:
            if not queue:
                return None
            index = queue.pop(0)
            nr = flags[index] if flags is not None else (random.randint(1, 10) == 1)   # worker.py:53
            return index, nr

    if groups == 1:
        _run_group(save_dir, network, size, visits, boards, seeds, device_index, next_game, stats, None, observer, lanes)
        if hasattr(network, "range_fallbacks"):             # forward passes redone in exact fp32 (f16 range guard)
            stats["range_fallbacks"] = network.range_fallbacks() - fb0
        return stats

    import torch
    sizes = [boards // groups + (1 if g < boards % groups else 0) for g in range(groups)]
    results = [dict(games=0, moves=0, leaf_evals=0) for _ in range(groups)]
    errors = []
    def work(g):
        try:
            stream = torch.cuda.Stream(device=torch.device("cuda", device_index))
            _run_group(save_dir, network, size, visits, sizes[g], seeds, device_index, next_game,
                       results[g], stream, None, lanes)
        except BaseException as exc:          # surfaced in the caller's thread
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(g,), name=f"selfplay-group-{g}") for g in range(groups)]
    # a group thread coming back from a (GIL-free) library call must not wait a whole default
    # switch interval (5 ms) for the interpreter while its GPU work queue runs dry
    import sys
    interval = sys.getswitchinterval()
    sys.setswitchinterval(2e-4)
    try:
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    finally:
        sys.setswitchinterval(interval)
    if errors:
        raise errors[0]
    for r in results:
        for k in ("games", "moves", "leaf_evals"):
            stats[k] += r[k]
    if hasattr(network, "range_fallbacks"):
        stats["range_fallbacks"] = network.range_fallbacks() - fb0
    return stats


def _auto_lanes(boards: int, size: int) -> int:
    """Lanes of a group of `boards` boards: one.  Lanes - several engines driven by ONE host thread through the two halves of a
    chained move - were measured (tools/experiments/sp_lanes_sweep*.sh, sp_pool_order.py; profiles/r06_selfplay_lanes_sweep.txt):
    two lanes of 8 boards run at 3.65-3.75 M leaf-evals/s or at 3.0-3.1 M - two stable states (in the slow one a lane's records
    are always ready at once and the other's 1.1 ms later; which one a run falls into went by its first calls, and neither a
    minimum distance between the lanes' begins nor forward caps moved it), three or four lanes collapse to 2.1-2.4 M.  Two
    groups on host threads of their own (selfplay_shard's `groups`) reach the fast state every time, so THAT is the default from
    8 to 28 boards; the lanes stay available (TG_SP_LANES / `lanes`) with the same games (tests/test_gpu_fastpath.py)."""
    env = os.environ.get("TG_SP_LANES")
    if env:
        return max(1, min(int(env), boards))
    return 1


def _run_lanes(save_dir, network, size, visits, lane_sizes, seeds, device_index, next_game, stats):
    """A group's boards as independent lanes on one host thread (selfplay_shard's `lanes`)."""
    import ctypes
    import time as _time
    from collections import deque
    import torch
    from tamago_amd import lib as _lib
    device = torch.device("cuda", device_index)
    start_board = GoBoard(board_size=size, komi=7.0, check_superko=True)
    komi = float(start_board.get_komi())
    a = size * size + 1

    class Lane:
        pass

    lanes = []
    timing = os.environ.get("TG_SP_TIMING") is not None
    t_end = t_begin = t_refill = 0.0
    n_moves = 0
    try:
        for n in lane_sizes:
            ln = Lane()
            ln.boards = n
            ln.engine = SearchEngine(size, n, max(SELF_PLAY_VISITS * 10, visits + 8), max(visits, 1),
                                     DeviceEvaluator(network), check_superko=True, device_index=device_index)
            ln.policy = torch.empty((n * ln.engine.K, a), dtype=torch.float32, device=device)
            ln.value = torch.empty((n * ln.engine.K, 3), dtype=torch.float32, device=device)
            if os.environ.get("TG_SP_LANE_STREAMS", "torch") == "own":
                ln.stream = ctypes.c_void_p()
                _lib.check(ln.engine.lib.tg_search_own_stream(ln.engine.handle, ctypes.byref(ln.stream)), "tg_search_own_stream")
            else:
                ln.torch_stream = torch.cuda.Stream(device=device)
                ln.stream = ctypes.c_void_p(ln.torch_stream.cuda_stream)
            ln.handle = ctypes.c_void_p()
            ln.sp_open = False
            lanes.append(ln)
            lib = ln.engine.lib
            _lib.check(lib.tg_selfplay_create(ln.engine.handle, os.fsencode(save_dir), visits, komi, repr(komi).encode(),
                                              ctypes.byref(ln.handle)), "tg_selfplay_create")
            ln.sp_open = True
            ln.finished = np.zeros(n, dtype=np.int32)
            ln.counts = np.zeros(3, dtype=np.int64)
            ln.live = 0
        lib = lanes[0].engine.lib

        def start(ln, slot: int) -> bool:
            nxt = next_game()
            if nxt is None:
                _lib.check(lib.tg_selfplay_start_game(ln.handle, slot, -1, 0), "tg_selfplay_start_game")
                return False
            index, never_resign = nxt
            ln.engine.streams[slot] = None
            ln.engine.set_root(slot, start_board, Stone.BLACK, np.random.RandomState(seeds[index]).get_state())
            _lib.check(lib.tg_selfplay_start_game(ln.handle, slot, index, int(never_resign)), "tg_selfplay_start_game")
            return True

        def begin(ln):
            _lib.check(lib.tg_selfplay_move_begin(ln.handle, network.handle, ln.engine.planes.data_ptr(), ln.policy.data_ptr(),
                                                  ln.value.data_ptr(), ln.stream), "tg_selfplay_move_begin")

        # slots are handed out lane by lane in board order - with one lane that is the order of the lock-step group
        for ln in lanes:
            for slot in range(ln.boards):
                if start(ln, slot):
                    ln.live += 1
                else:
                    ln.engine.set_root(slot, start_board, Stone.BLACK, np.random.RandomState(0).get_state())
        ring = deque()

        def begin_spaced(ln):
            begin(ln)

        for ln in lanes:
            ln.begun_at = 0.0
            if ln.live > 0:
                begin_spaced(ln)
                ring.append(ln)
        while ring:
            ln = ring.popleft()
            t0 = _time.perf_counter()
            _lib.check(lib.tg_selfplay_move_end(ln.handle, ln.finished.ctypes.data, ln.counts.ctypes.data), "tg_selfplay_move_end")
            t1 = _time.perf_counter()
            stats["games"] += int(ln.counts[0])
            stats["moves"] += int(ln.counts[1])
            stats["leaf_evals"] += int(ln.counts[2])
            for slot in np.nonzero(ln.finished)[0]:
                if not start(ln, int(slot)):
                    ln.live -= 1
            t2 = _time.perf_counter()
            if ln.live > 0:
                begin_spaced(ln)
                ring.append(ln)
            t3 = _time.perf_counter()
            t_end += t1 - t0
            t_refill += t2 - t1
            t_begin += t3 - t2
            n_moves += 1
        if timing:
            import sys
            sys.stderr.write(f"[selfplay timing] {len(lanes)} lanes, {n_moves} lane-moves: move_end {1e3 * t_end / max(n_moves, 1):.3f} ms, "
                             f"slot refill {1e3 * t_refill / max(n_moves, 1):.3f} ms, move_begin {1e3 * t_begin / max(n_moves, 1):.3f} ms per lane-move\n")
    finally:
        for ln in lanes:
            if getattr(ln, "sp_open", False):
                ln.engine.lib.tg_selfplay_destroy(ln.handle)
            if getattr(ln, "engine", None) is not None:
                ln.engine.close()


def _run_group(save_dir, network, size, visits, boards, seeds, device_index, next_game, stats, stream,
               observer=None, lanes=0):
    """One lock-step group of `boards` games on its own engine (and HIP stream, if given).

    With the library's own network the whole lock-step move is ONE library call (tg_selfplay_play_move:
    noise, halving schedule, every phase, the move decided and played on the device, the next root
    evaluated, records and SGF files on host threads); Python only starts games.  Any other evaluator is
    driven phase by phase from here (tg_selfplay_schedule / tg_selfplay_finish_move / tg_search_play)."""
    import contextlib
    import ctypes
    import torch
    from tamago_amd import lib as _lib
    from tamago_amd.nn.network.dual_net import DualNet
    if isinstance(network, DualNet) and observer is None:
        n_lanes = lanes if lanes > 0 else _auto_lanes(boards, size)
        n_lanes = max(1, min(n_lanes, boards))
        if n_lanes > 1 and (os.environ.get("TG_SP_CHAIN", "1") != "0"):
            sizes = [boards // n_lanes + (1 if g < boards % n_lanes else 0) for g in range(n_lanes)]
            _run_lanes(save_dir, network, size, visits, sizes, seeds, device_index, next_game, stats)
            return
    ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
    with ctx:
        evaluator = DeviceEvaluator(network) if isinstance(network, DualNet) \
            else HostEvaluator(network, torch.device("cuda", device_index))
        engine = SearchEngine(size, boards, max(SELF_PLAY_VISITS * 10, visits + 8), max(visits, 1),
                              evaluator, check_superko=True, device_index=device_index)
        lib = engine.lib
        start_board = GoBoard(board_size=size, komi=7.0, check_superko=True)
        handle = ctypes.c_void_p()
        _lib.check(lib.tg_selfplay_create(engine.handle, os.fsencode(save_dir), visits, float(start_board.get_komi()),
                                          repr(float(start_board.get_komi())).encode(), ctypes.byref(handle)),
                   "tg_selfplay_create")
        live = 0
        hook = None
        if observer is not None:
            if not isinstance(evaluator, DeviceEvaluator):
                raise ValueError("selfplay_shard: the observer taps tg_selfplay_play_move (DualNet evaluator only)")
            hook = _lib.SELFPLAY_OBSERVER(lambda _user, ev: observer(engine, ev.contents))
            _lib.check(lib.tg_selfplay_set_observer(handle, hook, None), "tg_selfplay_set_observer")

        def start(slot: int) -> bool:
            nxt = next_game()
            if nxt is None:
                _lib.check(lib.tg_selfplay_start_game(handle, slot, -1, 0), "tg_selfplay_start_game")
                return False
            index, never_resign = nxt
            engine.streams[slot] = None
            engine.set_root(slot, start_board, Stone.BLACK, np.random.RandomState(seeds[index]).get_state())
            _lib.check(lib.tg_selfplay_start_game(handle, slot, index, int(never_resign)), "tg_selfplay_start_game")
            return True

        try:
            for s in range(boards):
                if start(s):
                    live += 1
                else:
                    # fewer games than slots: park an empty board with a private stream
                    engine.set_root(s, start_board, Stone.BLACK, np.random.RandomState(0).get_state())
            finished = np.zeros(boards, dtype=np.int32)
            if isinstance(evaluator, DeviceEvaluator):
                # the library's own network: the whole move is one call (tg_selfplay_play_move)
                a = size * size + 1
                policy = torch.empty((boards * engine.K, a), dtype=torch.float32, device=engine.device)
                value = torch.empty((boards * engine.K, 3), dtype=torch.float32, device=engine.device)
                engine.sp_outputs = (policy, value)          # what an observer's device pointers refer to
                counts = np.zeros(3, dtype=np.int64)
                timing = os.environ.get("TG_SP_TIMING") is not None
                t_call = t_start = 0.0
                n_calls = 0
                import time as _time
                while live > 0:
                    t0 = _time.perf_counter()
                    _lib.check(lib.tg_selfplay_play_move(handle, network.handle, engine.planes.data_ptr(),
                                                         policy.data_ptr(), value.data_ptr(), engine._stream(),
                                                         finished.ctypes.data, counts.ctypes.data),
                               "tg_selfplay_play_move")
                    t1 = _time.perf_counter()
                    stats["games"] += int(counts[0])
                    stats["moves"] += int(counts[1])
                    stats["leaf_evals"] += int(counts[2])
                    for s in np.nonzero(finished)[0]:
                        if not start(int(s)):
                            live -= 1
                    t_call += t1 - t0
                    t_start += _time.perf_counter() - t1
                    n_calls += 1
                if timing:
                    import sys
                    sys.stderr.write(f"[selfplay timing] {n_calls} lock-step moves: play_move {1e3 * t_call / max(n_calls, 1):.3f} ms, "
                                     f"slot refill {1e3 * t_start / max(n_calls, 1):.3f} ms per move\n")
            else:
                # any other evaluator (host API): the phases are driven from here, the bookkeeping stays in C++
                max_phases = 16
                widths = np.zeros((max_phases, boards), dtype=np.int32)
                levels = np.zeros((max_phases, boards), dtype=np.int32)
                n_phases = ctypes.c_int32(0)
                played = np.zeros(boards, dtype=np.int32)
                counts = np.zeros(2, dtype=np.int64)
                while live > 0:
                    # boards are resident on the device (tg_search_play); parked slots keep their last root
                    engine.root_eval(use_logit=True)
                    engine.set_gumbel_noise()
                    _lib.check(lib.tg_selfplay_schedule(handle, widths.ctypes.data, levels.ctypes.data, max_phases,
                                                        ctypes.byref(n_phases)), "tg_selfplay_schedule")
                    stats["leaf_evals"] += live
                    for phase in range(n_phases.value):
                        engine.gumbel_phase(widths[phase], levels[phase])
                    stats["leaf_evals"] += int((widths[:n_phases.value].astype(np.int64) * levels[:n_phases.value]).sum())
                    _lib.check(lib.tg_selfplay_finish_move(handle, played.ctypes.data, finished.ctypes.data,
                                                           counts.ctypes.data), "tg_selfplay_finish_move")
                    stats["games"] += int(counts[0])
                    stats["moves"] += int(counts[1])
                    engine.play(played)
                    for s in np.nonzero(finished)[0]:
                        if not start(int(s)):
                            live -= 1
        finally:
            lib.tg_selfplay_destroy(handle)
            engine.close()


def selfplay_worker(save_dir: str, model_file_path: str, index_list: List[int], size: int,
                    visits: int, use_gpu: bool) -> None:
    """selfplay/worker.py:21-90.  One game at a time from ONE legacy stream seeded with
    ``random.choice(index_list)``, exactly like the reference worker."""
    from tamago_amd.nn.utility import load_network
    network = load_network(model_file_path=model_file_path, use_gpu=use_gpu, board_size=size)
    seed = random.choice(index_list)                                               # worker.py:39
    state = np.random.RandomState(seed).get_state()
    for index in index_list:
        if os.path.isfile(os.path.join(save_dir, f"{index}.sgf")):
            continue
        never_resign = random.randint(1, 10) == 1
        state = _play_one_game(save_dir, network, index, size, visits, state, never_resign)


def _play_one_game(save_dir, network, index, size, visits, rng_state, never_resign):
    """One game on a one-board engine, continuing `rng_state`; returns the stream state after
    the game so that the next game continues it (the reference seeds once per worker)."""
    from tamago_amd.mcts.time_manager import TimeManager, TimeControl
    from tamago_amd.mcts.tree import MCTSTree
    game = _Game(index, size, save_dir, never_resign)
    tree = MCTSTree(network, tree_size=max(SELF_PLAY_VISITS * 10, visits + 8))
    time_manager = TimeManager(TimeControl.CONSTANT_PLAYOUT, constant_visits=visits)
    saved = np.random.get_state()
    np.random.set_state(rng_state)
    try:
        winner, is_resign, score = Stone.EMPTY, False, 0.0
        for _ in range(size * size * 2):
            pos = tree.generate_move_with_sequential_halving(game.board, game.color, time_manager,
                                                             never_resign)
            if pos == RESIGN:
                winner, is_resign = Stone.get_opponent_color(game.color), True
                break
            game.board.put_stone(pos, game.color)
            game.pass_count = game.pass_count + 1 if pos == PASS else 0
            game.record.save_record(tree.get_root(), pos, game.color)
            game.color = Stone.get_opponent_color(game.color)
            if game.pass_count == 2:
                score = game.board.count_score() - game.board.get_komi()
                winner = Stone.BLACK if score > 0.1 else (Stone.WHITE if score < -0.1
                                                          else Stone.OUT_OF_BOARD)
                break
        _finish(game, winner, is_resign, score)
        return np.random.get_state()
    finally:
        np.random.set_state(saved)
