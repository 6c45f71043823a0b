#!/usr/bin/env python3
"""Headline benchmark: MCTS leaf-evaluations/s on 9x9, batch 256 (BASELINE.json).

Workload (config[1] of BASELINE.json): PUCT search, random-init
DualNet, 1000 strict visits per move, NN mini-batch 256 per tree -> per move and tree
1 root evaluation + mini-batches of 256/256/256/232 = 1001 leaf evaluations.  One
"step" = one such move search for every one of the `--trees` independent boards a GPU
drives in lock-step (self-play boards shard across boards and GPUs without any exchange,
SURVEY.md 8(e)); after each step every board plays its searched move, so tree shapes
vary like in self-play.  Inputs (root positions, weights) are resident before the timed
region; the timed region contains everything else, including the host side of the loop.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--trees T]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0) with the contract's keys plus `roofline` (fused forward
kernel: ALGORITHMIC FLOPs per launch / HIP-event launch time on the launch stream, against
the dense matrix peak of the precision the kernel executes in - SURVEY 8(d) -, the
issue-slot utilisation of the matrix pipe under its own name `mfma_issue_frac`, HBM traffic
from the rocprofv3 PMC summary measured on these kernel sources), `tree_kernels` (HBM traffic
of selection / backup, same rule), `cpu_baseline` (the CPU oracle - a port of the reference's
Python path - timed on this host on a bounded sample of the same workload) and legs for the
other BASELINE.json configs: at N = 1 `fp32_exact` (the same workload on the exact-fp32
Winograd kernel), `single_tree`, `cfg3_selfplay_16_boards`, `cfg4_shard_64_boards`,
`selfplay_1024_boards`, `cfg5_19x19`, `cfg5_19x19_trees` (with its own `roofline`), `cfg1_cpu`, `cpu_selfplay` (the
oracle's Gumbel worker as 4 / one-per-core OS processes); at N > 1 `cfg4_selfplay_shards` (one
64-board Gumbel shard per rank, aggregate + per-rank rates + host cores per rank).
"""
import argparse
import json
import os
import sys
import time

# (before the first HIP call of the process: the self-play legs drive up to ~8 HIP streams at once - tamago_amd/__init__.py)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# TG_BENCH_DRY_RUN=1 (tests/test_dist_shards.py): the N-rank PLUMBING of this file - process group, barriers, max-over-ranks
# clock, sum-over-ranks count, the cfg-4 gather with its failure path, the one JSON line - on CPU ranks (gloo) with a stub in
# place of the GPU workload, so that the 8-rank code path has run before an 8-GPU node shows up.  Its numbers mean nothing
# and the line says so ("data": "DRY RUN ...").
DRY = os.environ.get("TG_BENCH_DRY_RUN") == "1"

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 matrix (= fp32 vector) peak
PROFILES = os.path.join(REPO, "profiles")


def pmc_summary(kernel_name, size=9):
    """The rocprofv3 PMC summary (tools/pmc_r06.sh -> profiles/r06_pmc_forward_*.json; the scripts of earlier rounds are in the git history) measured on THESE
    kernel sources for THIS kernel: HBM-side bytes per position (FETCH_SIZE doubled + WRITE_SIZE, separate
    passes, as MI355X_MICROARCH.md section HBM prescribes), matrix-pipe busy fraction, L2 request bytes.
    A summary taken on other sources (csrc digest differs) or another kernel is NOT quoted: returns None and
    says so on stderr."""
    from tamago_amd.build import FORWARD_SOURCES, source_digest
    digest = source_digest(FORWARD_SOURCES)
    import glob
    # ("a + b": a forward pass of two kernels - the 19x19 tower + its batched heads kernel: the first one's summary, with the
    # HBM-side bytes of the others added)
    names = [n.split("<")[0].strip() for n in kernel_name.split(" + ")]
    found = {}
    for path in sorted(glob.glob(os.path.join(PROFILES, "r0[0-9]_pmc_forward_*.json")), reverse=True):
        with open(path) as f:
            d = json.load(f)
        if d.get("csrc_digest") != digest or f"_{size}x{size}_" not in os.path.basename(path):
            continue
        for n in names:
            # (a launch-size family: the largest profiled launch of the kernel stands for the throughput legs)
            if n in d.get("kernel", "") and (n not in found or d.get("positions_per_launch", 0) > found[n].get("positions_per_launch", 0)):
                d["file"] = os.path.relpath(path, REPO)
                found[n] = d
    if names[0] in found:
        d = found[names[0]]
        extra = [found[n] for n in names[1:] if n in found]
        if extra:
            d["derived"]["hbm_bytes_per_position"] += sum(e["derived"]["hbm_bytes_per_position"] for e in extra)
            d["file"] += " + " + " + ".join(e["file"] for e in extra)
        return d
    sys.stderr.write(f"bench.py: no PMC summary in profiles/ for kernel {kernel_name!r} at csrc digest {digest} - "
                     "roofline.traffic is null (run tools/pmc_r06.sh on the GPU box and commit the summary)\n")
    return None


def tree_pmc_summary():
    """HBM-side traffic of the tree kernels (selection, backup; rocprofv3 PMC, tools/pmc_r06.sh) - quoted only when the
    summary was measured on THESE sources (digest over all of csrc/ + the ABI header), else None + a warning."""
    import glob
    from tamago_amd.build import source_digest
    digest = source_digest()
    for path in sorted(glob.glob(os.path.join(PROFILES, "r0[0-9]_pmc_tree_and_featurize_kernels.json")), reverse=True):
        with open(path) as f:
            d = json.load(f)
        if d.get("csrc_digest") == digest:
            out = {"source": os.path.relpath(path, REPO)}
            for name, k in d.get("kernels", {}).items():
                if "hbm_bytes_per_leaf_eval" in k:
                    out[name] = {"hbm_bytes_per_leaf_eval": k["hbm_bytes_per_leaf_eval"], "hbm_GBps": k["hbm_GBps"],
                                 "fraction_of_8TBps_HBM_peak": k["fraction_of_8TBps_HBM_peak"],
                                 "avg_duration_us": k["avg_duration_us"]}
            return out
    sys.stderr.write(f"bench.py: no tree-kernel PMC summary in profiles/ at csrc digest {digest} - tree_kernels is null "
                     "(run tools/pmc_r06.sh on the GPU box and commit the summary)\n")
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--trees", type=int, default=2048, help="boards searched in lock-step per GPU")
    ap.add_argument("--groups", type=int, default=1,
                    help="run the boards of a GPU as this many lock-step groups on their own HIP streams "
                         "(measured: no gain - the forward kernel's two 256-VGPR waves per SIMD leave no "
                         "registers for another kernel's waves, so nothing overlaps: 2.25 / 2.20 / 2.08 M "
                         "leaf-evals/s at 1 / 2 / 4 groups)")
    ap.add_argument("--visits", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--size", type=int, default=9)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the extra legs (N = 1: exact-fp32 kernel, single tree, cfg-3 / cfg-4-shard self-play, cfg-5 "
                         "19x19, cfg-1 on the CPU; N > 1: the cfg-4 shard per rank)")
    ap.add_argument("--cfg4-boards", type=int, default=64, help="boards per cfg-4 shard (BASELINE.json: 64)")
    ap.add_argument("--cfg4-games", type=int, default=192, help="games each rank's cfg-4 shard plays to completion")
    ap.add_argument("--cfg4-visits", type=int, default=400)
    ap.add_argument("--cfg5-trees", type=int, default=256, help="19x19 trees of the cfg5_19x19_trees leg")
    ap.add_argument("--cpu-selfplay-seconds", type=float, default=6.0, help="per configuration of the cpu_selfplay leg")
    return ap.parse_args()


class TimedEvaluator:
    """DualNet forward on the launch stream, every launch bracketed by HIP events."""

    def __init__(self, network):
        self.network = network
        self.events = []
        self.batches = []
        self.record = False

    def __call__(self, planes, want_logits):
        if self.record:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self.network.forward_device(planes, want_logits)
            e1.record()
            self.events.append((e0, e1, planes.shape[0]))
        else:
            out = self.network.forward_device(planes, want_logits)
        self.batches.append(planes.shape[0])
        return out


def opening_boards(size, count, seed):
    """Synthetic root positions: seeded random legal openings of 0..11 plies."""
    from tamago_amd.board.go_board import GoBoard
    rs = np.random.RandomState(seed)
    boards, colors = [], []
    for i in range(count):
        board = GoBoard(size, 7.0, False)
        color = 1
        for _ in range(i % 12):
            while True:                                   # random empty point that is legal
                pos = board.onboard_pos[rs.randint(len(board.onboard_pos))]
                if board.is_legal(pos, color):
                    break
            board.put_stone(pos, color)
            color = 3 - color
        boards.append(board)
        colors.append(color)
    return boards, colors


def run_step(engines, plies_list, fresh_board, visits, batch):
    """One move search for every tree of every group; returns leaf evaluations done.  Boards live
    on the device: the searched move (arg-max visits) is played there (tg_search_play).  The
    groups are driven round-robin from this one host thread, each on its own stream: launches are
    asynchronous, only a group's own selection kernel is waited for (random-stream cursor)."""
    first = min(batch, visits)
    for engine, stream in engines:
        with torch.cuda.stream(stream):
            engine.root_eval(False, first_batch=first)
    done = 0
    while done < visits:
        k = min(batch, visits - done)
        for engine, stream in engines:
            with torch.cuda.stream(stream):
                engine.puct_batch(k)
        done += k
    leaves = 0
    for (engine, stream), plies in zip(engines, plies_list):
        with torch.cuda.stream(stream):
            # window for the next search, generated while this search's last forward pass runs
            engine.prefetch_rng(first)
            leaves += engine.T * (1 + visits)
            # -2: the most visited root child's move (get_best_move_index), chosen by the play kernel - the driver only
            # advances the positions and needs no read-back between two searches
            moves = np.full(engine.T, -2, dtype=np.int32)
            plies += 1
            over = plies > 2 * engine.P - 8
            moves[over] = -1
            engine.play(moves)
            for t in np.nonzero(over)[0]:                  # finished game: start a new one
                engine.set_root(int(t), fresh_board, 1)
                plies[t] = 0
    return leaves


def extra_legs(args, net, dev, local_rank, fresh_board):
    """Driver-visible numbers for the other BASELINE.json configs (never part of `value`): the literal one-tree
    cfg-2, cfg-3 (Gumbel self-play, 16 boards x 400 simulations), one cfg-4 shard (64 boards x 400 simulations;
    8 of them = cfg-4, one per GPU), a large self-play shard, and cfg-5 (19x19, 1600 visits, batch 64)."""
    import shutil
    import tempfile
    from tamago_amd.mcts.engine import SearchEngine
    from tamago_amd.nn.network.dual_net import DualNet
    from tamago_amd.selfplay.worker import selfplay_shard
    from tamago_amd import lib as _lib
    out = {}
    cur = torch.cuda.current_stream(dev)

    def one_tree(size, network, visits, batch, moves):
        board = type(fresh_board)(size, 7.0, False)
        one = SearchEngine(size, 1, visits + 16, batch, TimedEvaluator(network), device_index=local_rank)
        one.set_root(0, board, 1, np.random.RandomState(7).get_state())
        plies = np.zeros(1, dtype=np.int64)
        run_step([(one, cur)], [plies], board, visits, batch)
        torch.cuda.synchronize()
        # three timed runs of `moves` moves, the median reported (a single run of eight 10-ms moves spread 10 % between
        # otherwise identical bench runs; all three are in "runs_ms_per_move")
        runs = []
        for _ in range(3):
            t1 = time.perf_counter()
            n1 = sum(run_step([(one, cur)], [plies], board, visits, batch) for _ in range(moves))
            torch.cuda.synchronize()
            runs.append((time.perf_counter() - t1, n1))
        one.close()
        dt1, n1 = sorted(runs)[1]
        return {"value": n1 / dt1, "unit": "leaf-evals/s", "ms_per_move": dt1 / moves * 1e3, "moves": moves,
                "runs_ms_per_move": [round(d / moves * 1e3, 4) for d, _ in runs],
                "workload": f"ONE search tree, {size}x{size}, {visits} strict visits/move, NN batch {batch}",
                "forward_kernel": _lib.load().tg_net_kernel_name(network.handle, batch).decode()}

    # strict single-tree reading of config[1] (latency-bound: descent k+1 depends on the virtual loss of k)
    try:
        out["single_tree"] = one_tree(args.size, net, args.visits, args.batch, 8)
    except Exception as exc:                              # the headline must not depend on a leg
        out["single_tree"] = {"error": repr(exc)}

    def selfplay(boards, games):
        tmp = tempfile.mkdtemp(prefix="tg_sp_")
        try:
            ts = time.perf_counter()
            st = selfplay_shard(tmp, net, list(range(1, games + 1)), 9, 400, boards=boards,
                                never_resign_flags=[True] * games, device_index=local_rank)
            torch.cuda.synchronize()
            dts = time.perf_counter() - ts
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        return {"value": st["leaf_evals"] / dts, "unit": "leaf-evals/s", "games_per_hour": st["games"] / dts * 3600,
                "boards": boards, "games": st["games"], "moves": st["moves"], "seconds": dts,
                "workload": f"Gumbel sequential halving, 400 simulations/move, {boards} lock-step boards, "
                            "games to completion, SGF records written"}

    # games >> boards: a shard refills a slot when its game ends, but the run ends with the last game, and with
    # only two games per slot half of the lock-step moves ran on half-empty shards (9x9 games last 60..162 moves)
    for key, boards, games in (("cfg3_selfplay_16_boards", 16, 256), ("cfg4_shard_64_boards", 64, 640),
                               ("selfplay_1024_boards", 1024, 2048)):
        try:
            out[key] = selfplay(boards, games)
        except Exception as exc:                          # the headline must not depend on a leg
            out[key] = {"error": repr(exc)}
    net19 = None
    try:
        torch.manual_seed(4321)
        net19 = DualNet(dev, 19)
        out["cfg5_19x19"] = one_tree(19, net19, 1600, 64, 4)
    except Exception as exc:
        out["cfg5_19x19"] = {"error": repr(exc)}
    # config[4] as a THROUGHPUT workload: many 19x19 trees in lock-step, with the forward kernel's own roofline
    try:
        out["cfg5_19x19_trees"] = trees_19(net19 if net19 is not None else DualNet(dev, 19), local_rank, args.cfg5_trees)
    except Exception as exc:
        out["cfg5_19x19_trees"] = {"error": repr(exc)}
    return out


def trees_19(net19, local_rank, trees, visits=1600, batch=64, steps=2):
    """BASELINE.json config[4] (19x19, 1600 strict visits/move, NN batch 64) for `trees` boards in lock-step: leaf-evals/s and
    the roofline of the 19x19 forward kernel (algorithmic 322.5 MFLOP per position) from HIP events."""
    import ctypes
    from tamago_amd import lib as tl
    from tamago_amd.board.go_board import GoBoard
    from tamago_amd.mcts.engine import SearchEngine
    lib = tl.load()
    ev = TimedEvaluator(net19)
    cur = torch.cuda.current_stream()
    eng = SearchEngine(19, trees, visits + 16, batch, ev, device_index=local_rank)
    fresh = GoBoard(19, 7.0, False)
    for t in range(trees):
        eng.set_root(t, fresh, 1, np.random.RandomState(50_000 + t).get_state())
    plies = [np.zeros(trees, dtype=np.int64)]
    run_step([(eng, cur)], plies, fresh, visits, batch)                       # warm-up move
    ev.events.clear()
    ev.record = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    leaves = sum(run_step([(eng, cur)], plies, fresh, visits, batch) for _ in range(steps))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev.record = False
    full_b = trees * batch
    big = [e0.elapsed_time(e1) for e0, e1, b in ev.events if b == full_b]
    kern_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in ev.events)
    eng.close()
    avg_ms = float(np.mean(big)) if big else float("nan")
    flops_pos = lib.tg_net_flops_per_position(19)
    peak = ctypes.c_double(0.0)
    dtype_name = ctypes.c_char_p()
    exec_flops = lib.tg_net_executed_flops_per_position(net19.handle, full_b, ctypes.byref(peak), ctypes.byref(dtype_name))
    algorithmic = full_b * flops_pos / (avg_ms * 1e-3) / 1e12
    kname19 = lib.tg_net_kernel_name(net19.handle, full_b).decode()
    pmc = pmc_summary(kname19, 19)
    return {"value": leaves / dt, "unit": "leaf-evals/s", "trees": trees, "steps": steps, "ms_per_step": dt / steps * 1e3,
            "workload": f"cfg-5: {trees} lock-step 19x19 trees, PUCT, {visits} strict visits/move, NN batch {batch} per tree",
            "roofline": {"bound": "mfma", "kernel": kname19,
                         "achieved": algorithmic, "peak": peak.value, "unit": "TFLOP/s", "frac": algorithmic / peak.value,
                         "algorithmic_flops_per_position": flops_pos,
                         "mfma_issue_tflops": full_b * exec_flops / (avg_ms * 1e-3) / 1e12,
                         "mfma_issue_frac": full_b * exec_flops / (avg_ms * 1e-3) / 1e12 / peak.value,
                         "executed_dtype": dtype_name.value.decode() if dtype_name.value else "",
                         "avg_launch_ms": avg_ms, "launches": len(big), "positions_per_launch": full_b,
                         "forward_share_of_step": kern_ms * 1e-3 / dt,
                         "traffic": pmc["derived"]["hbm_bytes_per_position"] * full_b if pmc else None,
                         "traffic_unit": "HBM-side bytes per launch: rocprofv3 PMC FETCH_SIZE x 2 + WRITE_SIZE (separate passes) "
                                         "per position x positions of this launch",
                         "traffic_source": pmc["file"] if pmc else None,
                         "algorithmic_io_bytes": full_b * (6 * 361 * 4 + (361 + 4) * 4),
                         "mfma_busy_frac_pmc": pmc["derived"].get("mfma_busy_fraction_of_simd_cycles") if pmc else None}}


def cpu_baseline(size, visits, batch, budget_s):
    """The CPU oracle (port of the reference's Python/NumPy tree + PyTorch-CPU DualNet)
    on the same workload, bounded to ~budget_s seconds."""
    from oracle.board import GoBoard as OBoard, BLACK
    from oracle.net import OracleNet
    from oracle.tree import MCTSTree as OTree, TimeManager as OTM, TimeControl as OTC
    from tamago_amd.nn.network.dual_net import random_state_dict
    torch.manual_seed(0)
    net = OracleNet(random_state_dict(size))
    tree = OTree(net, size, tree_size=visits + 16, batch_size=batch)
    board = OBoard(size)
    np.random.seed(0)
    color = BLACK
    tm = OTM(OTC.STRICT_PLAYOUT, visits)
    mv = tree.search_best_move(board, color, tm)            # warm-up move
    board.put_stone(mv if mv > 0 else 0, color)
    color = 3 - color
    t0 = time.time()
    leaves = 0
    moves = 0
    while time.time() - t0 < budget_s and moves < 30:
        mv = tree.search_best_move(board, color, tm)
        leaves += sum(tree.batch_log)
        tree.batch_log.clear()
        board.put_stone(mv if mv > 0 else 0, color)
        color = 3 - color
        moves += 1
    dt = time.time() - t0
    return {"value": leaves / dt, "unit": "leaf-evals/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{moves} moves x {visits + 1} leaf-evals, 1 tree, batch {batch}, "
                      f"oracle Python tree (1 thread) + PyTorch-CPU DualNet "
                      f"({torch.get_num_threads()} threads) of {os.cpu_count()} host cores, "
                      f"{dt:.1f} s"}


def cpu_selfplay(seconds):
    """the CPU comparison for the self-play configs is N worker PROCESSES (selfplay_main.py:58-65; reference default
    NUM_SELF_PLAY_WORKERS = 4, learning_param.py:43; also one per core): the oracle's Gumbel worker (oracle/cpu_selfplay.py) as
    4 processes and as one process per core (capped at 32), at 16 and at 400 simulations per move, `seconds` each; torch
    threads per process = cores / processes.  Leaf evaluations of all workers / the longest worker's time."""
    import subprocess
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    out = {"kind": "port", "host_cores": cores,
           "workload": "Gumbel sequential-halving self-play on the CPU oracle (Python tree + PyTorch-CPU DualNet), one OS process "
                       "per worker as selfplay_main.py starts them, games to completion"}
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), CUDA_VISIBLE_DEVICES="",
               HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS="1")
    for procs in sorted({4, max(1, min(cores, 32))}):
        threads = max(1, cores // procs)
        for visits in (16, 400):
            env["OMP_NUM_THREADS"] = str(threads)
            ps = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_selfplay", "--seconds", str(seconds), "--visits", str(visits),
                                    "--seed", str(1 + i), "--threads", str(threads)], env=env, cwd=REPO,
                                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(procs)]
            rows = []
            for pr in ps:
                try:
                    so, _ = pr.communicate(timeout=seconds * 4 + 240)
                    rows.append(json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1]))
                except Exception:
                    pr.kill()
            key = f"{procs}_processes_{visits}_sims"
            if len(rows) != procs:
                out[key] = {"error": f"{procs - len(rows)} of {procs} workers gave no result"}
                continue
            total = sum(r["leaf_evals"] for r in rows)
            slowest = max(r["seconds"] for r in rows)
            out[key] = {"value": total / slowest, "unit": "leaf-evals/s", "processes": procs, "torch_threads_per_process": threads,
                        "cores": min(cores, procs * threads), "simulations_per_move": visits, "seconds": slowest,
                        "moves": sum(r["moves"] for r in rows)}
    return out


def cfg1_cpu(budget_s):
    """BASELINE.json config[0]: 9x9, random-init DualNet, 100 visits/move, NN batch 1, CPU path only (the reference's
    plumbing: main.py --visits 100 --batch-size 1 --use-gpu false): the CPU oracle, STRICT_PLAYOUT (exactly 100 visits)
    and CONSTANT_PLAYOUT (the reference's --visits: early stop once the runner-up cannot catch up)."""
    from oracle.board import GoBoard as OBoard, BLACK
    from oracle.net import OracleNet
    from oracle.tree import MCTSTree as OTree, TimeManager as OTM, TimeControl as OTC
    from tamago_amd.nn.network.dual_net import random_state_dict
    torch.manual_seed(0)
    net = OracleNet(random_state_dict(9))
    # batch-1 convolutions do not scale over a big host's cores (128 torch threads: 30 leaf-evals/s on the GPU box, 8
    # threads: the survey container's 150-200): the leg runs at 8 torch threads and says so
    all_threads = torch.get_num_threads()
    torch.set_num_threads(min(8, all_threads))
    out = {"kind": "port", "cores": torch.get_num_threads(),
           "workload": "cfg-1: 9x9, 100 visits/move, NN batch 1, CPU oracle (Python tree, 1 thread + PyTorch-CPU DualNet)"}
    for label, mode in (("strict", OTC.STRICT_PLAYOUT), ("constant", OTC.CONSTANT_PLAYOUT)):
        tree = OTree(net, 9, tree_size=256, batch_size=1)
        board = OBoard(9)
        np.random.seed(0)
        color = BLACK
        tm = OTM(mode, 100)
        t0 = time.time()
        leaves = moves = 0
        while time.time() - t0 < budget_s / 2 and moves < 12:
            mv = tree.search_best_move(board, color, tm)
            leaves += sum(tree.batch_log)
            tree.batch_log.clear()
            board.put_stone(mv if mv > 0 else 0, color)
            color = 3 - color
            moves += 1
        dt = time.time() - t0
        out[label] = {"value": leaves / dt, "unit": "leaf-evals/s", "moves": moves, "leaf_evals": leaves,
                      "ms_per_move": dt / max(moves, 1) * 1e3}
    torch.set_num_threads(all_threads)
    return out


def cfg4_leg(args, net, local_rank, rank, world, dist, red_dev):
    """BASELINE.json config[3] as the driver's N-rank launch runs it: every rank = one GPU = ONE Gumbel self-play shard of
    `--cfg4-boards` lock-step boards at 400 simulations (selfplay/worker.py via selfplay_shard, the launcher's code path,
    selfplay_main.py:44-65), no collective in the data path; rank 0 reports the aggregate over the ranks (leaf evaluations
    of all shards / the slowest shard's time), the per-rank rates and the host cores each rank was pinned to."""
    import shutil
    import tempfile
    from tamago_amd.selfplay.worker import selfplay_shard
    games = args.cfg4_games
    first = 1 + rank * games
    tmp = tempfile.mkdtemp(prefix=f"tg_cfg4_r{rank}_")
    if not DRY:
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    cpu0 = time.process_time()                              # CPU seconds of ALL threads of this rank (group threads, the library's random-stream generators)
    # A rank whose shard fails still takes part in every collective below: it contributes an error row (the other
    # ranks would otherwise wait in all_gather until the RCCL timeout).  TG_BENCH_FAIL_RANK=r injects such a failure (tests).
    err = None
    st = {"leaf_evals": 0, "games": 0, "moves": 0}
    try:
        if os.environ.get("TG_BENCH_FAIL_RANK") == str(rank):
            raise RuntimeError(f"injected failure on rank {rank} (TG_BENCH_FAIL_RANK)")
        if DRY:
            time.sleep(0.05 * (1 + rank % 3))
            st = {"leaf_evals": (args.cfg4_visits + 1) * 60 * games, "games": games, "moves": 60 * games}
        else:
            st = selfplay_shard(tmp, net, list(range(first, first + games)), 9, args.cfg4_visits, boards=args.cfg4_boards,
                                never_resign_flags=[True] * games, device_index=local_rank)
            torch.cuda.synchronize()
    except Exception as exc:                                # (KeyboardInterrupt / SystemExit end the process: torchrun
        err = repr(exc)                                     #  then tears the other ranks down)
        sys.stderr.write(f"bench.py rank {rank}: cfg-4 shard failed: {err}\n")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    dt = time.perf_counter() - t0
    cpu_s = time.process_time() - cpu0
    mine = torch.tensor([st["leaf_evals"], dt, st["games"], st["moves"], len(os.sched_getaffinity(0)), 1.0 if err else 0.0, cpu_s],
                        dtype=torch.float64, device=red_dev)
    if world > 1:
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
    else:
        allr = [mine]
    rows = [[float(v) for v in t.cpu()] for t in allr]
    failed = [i for i, r in enumerate(rows) if r[5] != 0.0]
    if failed:
        # every rank sees the same flags, so every rank enters this second collective
        msgs = [None] * world
        if world > 1:
            dist.all_gather_object(msgs, err)
        else:
            msgs = [err]
        return {"error": f"cfg-4 shard failed on rank(s) {failed}", "failed_ranks": failed,
                "messages": {str(i): msgs[i] for i in failed}, "shards": world}
    total = sum(r[0] for r in rows)
    slowest = max(r[1] for r in rows)
    return {"value": total / slowest, "unit": "leaf-evals/s", "shards": world, "boards_per_shard": args.cfg4_boards,
            "games_per_shard": games, "simulations_per_move": args.cfg4_visits, "seconds": slowest,
            "games_per_hour": sum(r[2] for r in rows) / slowest * 3600,
            # host cost of a shard: CPU seconds of all its threads per 10^6 leaf evaluations, and how many cores that keeps
            # busy while the shard runs - host contention is the only thing that can bend the 1 -> 8 GPU curve
            "host_cpu_s_per_1e6_leaf_evals": sum(r[6] for r in rows) / max(total, 1.0) * 1e6,
            "per_rank": [{"rank": i, "leaf_evals_per_s": r[0] / r[1], "seconds": r[1], "games": int(r[2]), "moves": int(r[3]),
                          "host_cores": int(r[4]), "host_cpu_s": r[6], "host_cores_busy": r[6] / r[1]} for i, r in enumerate(rows)],
            "workload": f"cfg-4: {world} shard(s) x {args.cfg4_boards} lock-step boards, Gumbel sequential halving, "
                        f"{args.cfg4_visits} simulations/move, games to completion, SGF records written, one shard per rank / GPU"}


def dry_main(args, rank, local_rank, world):
    """TG_BENCH_DRY_RUN=1: main()'s rank plumbing with a stub workload on CPU ranks (gloo).  Same collectives in the same
    order as the GPU path: barrier / timed steps / barrier, MAX over ranks of the clock, SUM of the counts, the host-cost
    gather, the cfg-4 leg's gather with its failure path, one JSON line from rank 0, barrier, teardown."""
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        try:
            from tamago_amd.selfplay.main import pin_host_threads
            pin_host_threads(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), None)
        except Exception as exc:
            sys.stderr.write(f"bench.py: host-thread pinning skipped ({exc})\n")
    red_dev = torch.device("cpu")

    def barrier():
        if world > 1:
            dist.barrier()

    def step():
        time.sleep(0.01 * (1 + rank % 3))                      # ranks finish at different times: the clock is the slowest one's
        return args.trees * (args.visits + 1)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    cpu0 = time.process_time()
    leaves = 0
    for _ in range(args.steps):
        leaves += step()
    barrier()
    elapsed = time.perf_counter() - t0
    row = [time.process_time() - cpu0, elapsed, float(leaves), float(len(os.sched_getaffinity(0)))]
    rows = [row]
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([leaves], dtype=torch.float64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        leaves = float(tot.item())
        mine_h = torch.tensor(row, dtype=torch.float64)
        all_h = [torch.zeros_like(mine_h) for _ in range(world)]
        dist.all_gather(all_h, mine_h)
        rows = [[float(v) for v in t] for t in all_h]
    result = None
    if rank == 0:
        result = {"metric": "MCTS leaf-evals/sec (9x9, batch 256)", "value": leaves / elapsed, "unit": "leaf-evals/s",
                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                  "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none",
                  "data": "DRY RUN (TG_BENCH_DRY_RUN=1): no GPU, stub workload - rank plumbing only, the numbers mean nothing",
                  "config": {"workload": "stub", "trees_per_gpu": args.trees,
                             "leaf_evals_per_step_per_gpu": args.trees * (args.visits + 1),
                             "parallelism": f"{world} x independent board shards (no collective)"},
                  "host": {"per_rank": [{"rank": i, "cpu_s": r[0], "cores_busy": r[0] / r[1], "cores_pinned": int(r[3])}
                                        for i, r in enumerate(rows)]}}
    if world > 1 and not args.no_legs:
        try:
            leg = cfg4_leg(args, None, local_rank, rank, world, dist, red_dev)
        except Exception as exc:
            leg = {"error": repr(exc)}
        if rank == 0:
            result["cfg4_selfplay_shards"] = leg
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("TG_SINGLE_DEVICE"):                     # N ranks on one GPU (tests only)
        local_rank = 0
    if DRY:
        return dry_main(args, rank, local_rank, world)
    torch.cuda.set_device(local_rank)                          # before any collective is set up
    dev = torch.device("cuda", local_rank)
    backend = os.environ.get("TG_DIST_BACKEND", "nccl")        # nccl == RCCL on ROCm
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        # one rank per GPU shares the host with the others: give each its own slice of the cores (the library's
        # random-stream threads of eight ranks must not pile onto the same cores)
        try:
            from tamago_amd.selfplay.main import pin_host_threads
            pin_host_threads(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), local_rank)
        except Exception as exc:                                  # pinning is an optimisation, never fatal
            sys.stderr.write(f"bench.py: host-thread pinning skipped ({exc})\n")

    from tamago_amd import lib as tl
    from tamago_amd.mcts.engine import SearchEngine
    from tamago_amd.nn.network.dual_net import DualNet
    lib = tl.load()

    torch.manual_seed(1234)
    net = DualNet(dev, args.size)                      # random-init weights, resident
    evaluator = TimedEvaluator(net)
    n_groups = max(1, min(args.groups, args.trees))
    sizes = [args.trees // n_groups + (1 if g < args.trees % n_groups else 0) for g in range(n_groups)]
    boards, colors = opening_boards(args.size, args.trees, 1000 + rank)
    engines, plies_list = [], []
    t0 = 0
    for g, n in enumerate(sizes):
        stream = torch.cuda.Stream(device=dev) if n_groups > 1 else torch.cuda.current_stream(dev)
        with torch.cuda.stream(stream):
            engine = SearchEngine(args.size, n, args.visits + 16, args.batch, evaluator,
                                  device_index=local_rank)
            for t in range(n):                          # one private legacy stream per board
                rs = np.random.RandomState(10_000 * rank + t0 + t)
                engine.set_root(t, boards[t0 + t], colors[t0 + t], rs.get_state())
        engines.append((engine, stream))
        plies_list.append(np.array([b.moves - 1 for b in boards[t0:t0 + n]], dtype=np.int64))
        t0 += n

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    from tamago_amd.board.go_board import GoBoard
    import ctypes
    fresh_board = GoBoard(args.size, 7.0, False)
    full_b = sizes[0] * args.batch
    flops_pos = lib.tg_net_flops_per_position(args.size)

    host_cpu = []                                             # (CPU seconds of all threads, wall seconds, leaves) of this rank's timed regions
    host_threads = []                                         # per timed region: [(thread name, CPU seconds)] of the busiest threads

    def thread_cpu():
        """{tid: (comm, CPU seconds)} of this process's threads (/proc/self/task/*/stat: utime + stime)."""
        out = {}
        tick = os.sysconf("SC_CLK_TCK")
        try:
            for tid in os.listdir("/proc/self/task"):
                with open(f"/proc/self/task/{tid}/stat") as f:
                    text = f.read()
                comm = text[text.index("(") + 1:text.rindex(")")]
                rest = text[text.rindex(")") + 2:].split()
                out[tid] = (comm, (int(rest[11]) + int(rest[12])) / tick)
        except OSError:
            pass
        return out

    def timed_region(steps, warmup):
        """W untimed + K timed steps (barrier + synchronise on both sides); returns leaf evaluations, seconds (max over
        ranks), and the forward launches' HIP-event times (all: ms sum; full-size launches: list)."""
        evaluator.record = False
        for _ in range(warmup):
            run_step(engines, plies_list, fresh_board, args.visits, args.batch)
        evaluator.events.clear()
        evaluator.record = True
        barrier()
        t0 = time.perf_counter()
        cpu0 = time.process_time()
        thr0 = thread_cpu()
        leaves = 0
        for _ in range(steps):
            leaves += run_step(engines, plies_list, fresh_board, args.visits, args.batch)
        barrier()
        elapsed = time.perf_counter() - t0
        host_cpu.append((time.process_time() - cpu0, elapsed, leaves))
        thr1 = thread_cpu()
        busy = sorted(((c, s1 - thr0.get(tid, (c, 0.0))[1]) for tid, (c, s1) in thr1.items()), key=lambda kv: -kv[1])
        host_threads.append([(c, round(v, 3)) for c, v in busy[:6] if v > 0.0])
        evaluator.record = False
        kern_ms, big = 0.0, []
        for e0, e1, b in evaluator.events:
            ms = e0.elapsed_time(e1)
            kern_ms += ms
            if b == full_b:
                big.append(ms)
        if world > 1:
            tmax = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
            tot = torch.tensor([leaves], device=red_dev, dtype=torch.float64)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            leaves = float(tot.item())
        return leaves, elapsed, kern_ms, big

    def roofline(leaves, elapsed, kern_ms, big):
        """achieved = ALGORITHMIC FLOPs per launch (72.28 MFLOP per 9x9 position x positions) / the
        kernel's average launch time (HIP events on the launch stream), against the dense matrix peak of the precision
        the kernel executes in.  mfma_issue_frac = the FLOPs of the MFMA instructions it issues (operand splitting, tile
        padding) over the same peak: the issue-slot utilisation of the matrix pipe."""
        avg_ms = float(np.mean(big)) if big else float("nan")
        kname = lib.tg_net_kernel_name(net.handle, full_b).decode()
        peak = ctypes.c_double(0.0)
        dtype_name = ctypes.c_char_p()
        exec_flops = lib.tg_net_executed_flops_per_position(net.handle, full_b, ctypes.byref(peak), ctypes.byref(dtype_name))
        algorithmic = full_b * flops_pos / (avg_ms * 1e-3) / 1e12
        executed = full_b * exec_flops / (avg_ms * 1e-3) / 1e12
        pmc = pmc_summary(kname) if args.size == 9 else None
        io_bytes = 6 * args.size ** 2 * 4 + (args.size ** 2 + 4) * 4
        return peak.value, {
            "bound": "mfma",
            "kernel": kname,
            "note": "achieved = algorithmic FLOPs (direct 3x3 convolution count of SURVEY 3.4: 72.28 MFLOP per 9x9 position) "
                    "per launch / HIP-event launch time; peak = dense matrix peak of the precision the kernel EXECUTES in "
                    "(f16 pipe: 2 500, fp32 pipe: 157.3 TFLOP/s); frac = achieved / peak.  mfma_issue_* = FLOPs of the MFMA "
                    "instructions actually issued (three f16 products per fp32 product, tile padding) over the same peak = "
                    "issue-slot utilisation of the matrix pipe.  The board is power-capped: see power_cap.",
            "achieved": algorithmic,
            "peak": peak.value,
            "unit": "TFLOP/s",
            "frac": algorithmic / peak.value,
            "algorithmic_flops_per_position": flops_pos,
            "algorithmic_over_fp32_matrix_peak": algorithmic / FP32_MFMA_PEAK_TFLOPS,
            "mfma_issue_tflops": executed,
            "mfma_issue_frac": executed / peak.value,
            "executed_dtype": dtype_name.value.decode() if dtype_name.value else "",
            "executed_flops_per_position": exec_flops,
            "power_cap": "MI355X throttles the shader clock to its 1.4 kW budget: back-to-back v_mfma_f32_16x16x32_f16 on random "
                         "data from registers sustain 1 930 TFLOP/s at 1.88 GHz, not 2 500 (profiles/r03_microbench_mfma_power.txt)",
            "mfma_issue_frac_of_power_capped_rate": executed / 1930.0 if peak.value > 200 else None,
            "traffic": pmc["derived"]["hbm_bytes_per_position"] * full_b if pmc else None,
            "traffic_unit": "HBM-side bytes per launch: rocprofv3 PMC FETCH_SIZE x 2 + WRITE_SIZE (separate passes) "
                            "per position x positions of this launch",
            "traffic_source": pmc["file"] if pmc else None,
            "algorithmic_io_bytes": full_b * io_bytes,
            "mfma_busy_frac_pmc": pmc["derived"]["mfma_busy_fraction_of_simd_cycles"] if pmc else None,
            "avg_launch_ms": avg_ms,
            "launches": len(big),
            "positions_per_launch": full_b,
            "forward_share_of_step": kern_ms * 1e-3 / elapsed,
            "end_to_end_frac": leaves / elapsed / world * flops_pos / 1e12 / peak.value,
        }

    leaves, elapsed, kern_ms, big = timed_region(args.steps, args.warmup)
    # host cost of the headline loop, every rank's: CPU seconds of all its threads (this driver thread + the library's
    # random-stream generator threads) per 10^6 leaf evaluations, cores kept busy
    cpu_s, wall_s, my_leaves = host_cpu[0]
    host_rows = [[cpu_s, wall_s, float(my_leaves), float(len(os.sched_getaffinity(0)))]]
    if world > 1:
        mine_h = torch.tensor(host_rows[0], dtype=torch.float64, device=red_dev)
        all_h = [torch.zeros_like(mine_h) for _ in range(world)]
        dist.all_gather(all_h, mine_h)
        host_rows = [[float(v) for v in t.cpu()] for t in all_h]

    result = None
    if rank == 0:
        peak_value, roof = roofline(leaves, elapsed, kern_ms, big)
        result = {
            "metric": "MCTS leaf-evals/sec (9x9, batch 256)" if args.size == 9
            else f"MCTS leaf-evals/sec ({args.size}x{args.size}, batch {args.batch})",
            "value": leaves / elapsed,
            "unit": "leaf-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if peak_value < 200 else "f32 results from f16 x 2 split operands, fp32 accumulate",
            "data": "synthetic",
            "config": {
                "workload": f"cfg-2 PUCT {args.size}x{args.size}, random-init DualNet, "
                            f"{args.visits} strict visits/move, NN batch {args.batch} per tree",
                "trees_per_gpu": args.trees,
                "lockstep_groups_per_gpu": n_groups,
                "leaf_evals_per_step_per_gpu": args.trees * (args.visits + 1),
                "parallelism": f"{world} x independent board shards (no collective)",
            },
            "roofline": roof,
            "host": {"cpu_s_per_1e6_leaf_evals": sum(r[0] for r in host_rows) / max(sum(r[2] for r in host_rows), 1.0) * 1e6,
                     "per_rank": [{"rank": i, "cpu_s": r[0], "cores_busy": r[0] / r[1], "cores_pinned": int(r[3])}
                                  for i, r in enumerate(host_rows)],
                     "busiest_threads_cpu_s": host_threads[0] if host_threads else None,
                     "note": "time.process_time() of each rank over its timed region: all threads of the process; cores_busy = "
                             "CPU seconds / wall seconds.  Since round 6 the random streams are generated on the device (no "
                             "generator threads); busiest_threads_cpu_s = rank 0's threads by CPU seconds (/proc/self/task)"},
            "tree_kernels": tree_pmc_summary() if args.size == 9 else None,
            # forward launches the exact-fp32 kernel had to redo (f16 range guard of the split-operand kernels): 0 = none
            "range_fallbacks": net.range_fallbacks(),
        }
    # ---- exact-fp32 arithmetic on the same workload (N = 1): the Winograd fp32-MFMA kernel ----
    if world == 1 and not args.no_legs and args.size == 9:
        saved = os.environ.get("TG_FWD_ALGO")
        os.environ["TG_FWD_ALGO"] = "wino"
        try:
            l2, e2, k2, b2 = timed_region(2, 1)
            _, roof2 = roofline(l2, e2, k2, b2)
            result["fp32_exact"] = {"value": l2 / e2, "unit": "leaf-evals/s", "dtype": "f32", "steps": 2, "warmup": 1,
                                    "ms_per_step": e2 / 2 * 1e3, "workload": result["config"]["workload"] +
                                    f", {args.trees} trees, TG_FWD_ALGO=wino (exact fp32 operands on the fp32 MFMA)",
                                    "roofline": roof2,
                                    "note": "fallback path only (the exact redo behind the f16 range guard; rare: range_fallbacks), "
                                            "not tuned: its HBM-side traffic is ~24x the algorithmic bytes and half of its LDS-active "
                                            "cycles are bank conflicts (profiles/r05_pmc_forward_wino_9x9_b65536.json).  Its "
                                            "roofline.frac divides the DIRECT-convolution FLOP count (72.28 MFLOP per position) by the "
                                            "fp32 matrix peak although the kernel is a Winograd F(2x2,3x3) tower that issues 2.25x fewer "
                                            "products - a frac above 1 is therefore possible and says nothing about utilisation; "
                                            "mfma_issue_frac (issued MFMA FLOPs over the same peak) is the utilisation figure"}
        except Exception as exc:                          # the headline must not depend on a leg
            result["fp32_exact"] = {"error": repr(exc)}
        finally:
            if saved is None:
                os.environ.pop("TG_FWD_ALGO", None)
            else:
                os.environ["TG_FWD_ALGO"] = saved
    # ---- BASELINE.json config[3]: one Gumbel self-play shard per rank (N > 1: what the driver's scaling run adds) ----
    if world > 1 and not args.no_legs and args.size == 9:
        for eng, _ in engines:
            eng.close()
        engines = []
        try:
            leg = cfg4_leg(args, net, local_rank, rank, world, dist, red_dev)
        except Exception as exc:
            # every rank must reach the collectives below whatever happened here
            leg = {"error": repr(exc)}
            sys.stderr.write(f"bench.py rank {rank}: cfg-4 leg failed: {exc!r}\n")
        if rank == 0:
            result["cfg4_selfplay_shards"] = leg
    if rank == 0:
        if world == 1 and not args.no_legs and args.size == 9:
            for eng, _ in engines:
                eng.close()
            torch.cuda.synchronize()
            result.update(extra_legs(args, net, dev, local_rank, fresh_board))
        if not args.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only
            try:
                result["cpu_baseline"] = cpu_baseline(args.size, args.visits, args.batch, args.cpu_seconds)
            except Exception as exc:                          # a broken baseline leg must not lose the measured line
                result["cpu_baseline"] = {"error": repr(exc)}
            if not args.no_legs and args.size == 9:
                try:
                    result["cfg1_cpu"] = cfg1_cpu(8.0)
                except Exception as exc:
                    result["cfg1_cpu"] = {"error": repr(exc)}
                try:
                    result["cpu_selfplay"] = cpu_selfplay(args.cpu_selfplay_seconds)
                except Exception as exc:
                    result["cpu_selfplay"] = {"error": repr(exc)}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
