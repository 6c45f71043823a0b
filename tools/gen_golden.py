#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Only works in the build container, where the read-only reference checkout lives at
/root/reference.  The reference is imported (never copied): for 19x19 a scratch copy
with ``BOARD_SIZE = 19`` is made under /tmp at run time (the board size is a module
constant, board/constant.py:4) and deleted afterwards.  The fixtures are pure data
(inputs + expected outputs); this script is committed so they can be regenerated.

    python tools/gen_golden.py            # both sizes
    python tools/gen_golden.py --size 9   # worker mode (PYTHONPATH must point at a reference tree)
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")


def orchestrate():
    if not os.path.isdir(REF):
        print("no reference checkout at", REF, "- nothing to do")
        return 0
    os.makedirs(GOLD, exist_ok=True)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env["PYTHONPATH"] = REF + os.pathsep + REPO
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--size", "9"], env=env,
                          cwd="/tmp")
    for size in (19, 13):                  # (13: round 6 - the third size the library is built for)
        scratch = tempfile.mkdtemp(prefix=f"ref{size}_")
        try:
            tree = os.path.join(scratch, "ref")
            shutil.copytree(REF, tree, ignore=shutil.ignore_patterns(".git", "__pycache__"))
            path = os.path.join(tree, "board", "constant.py")
            text = open(path, encoding="utf-8").read().replace("BOARD_SIZE = 9", f"BOARD_SIZE = {size}")
            open(path, "w", encoding="utf-8").write(text)
            env["PYTHONPATH"] = tree + os.pathsep + REPO
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--size", str(size)],
                                  env=env, cwd="/tmp")
        finally:
            shutil.rmtree(scratch, ignore_errors=True)
    return 0


# ======================================================================================
# worker: runs with the reference on sys.path
# ======================================================================================
def worker(size: int):
    import random
    import numpy as np
    import torch

    from board.constant import BOARD_SIZE, PASS
    assert BOARD_SIZE == size, (BOARD_SIZE, size)
    from board.go_board import GoBoard
    from board.stone import Stone
    from board.pattern import Pattern
    from nn.feature import generate_input_planes
    from nn.network.dual_net import DualNet
    from mcts.tree import MCTSTree
    from mcts.time_manager import TimeManager, TimeControl
    from mcts.sequential_halving import get_candidates_and_visit_pairs

    from oracle.stubnet import StubNet
    from oracle.net import make_state_dict

    torch.set_grad_enabled(False)
    P = size * size
    tag = f"s{size}"

    def col(c):
        return Stone.BLACK if c == 1 else Stone.WHITE

    # ---------------------------------------------------------------- tables
    if size == 9:
        pat = Pattern(size, lambda x, y: x + y * (size + 2))
        eye = np.array([e.value for e in pat.eye], dtype=np.uint8)
        tables = {
            "eye_black_codes": [int(i) for i in np.nonzero(eye == 1)[0]],
            "eye_white_codes": [int(i) for i in np.nonzero(eye == 2)[0]],
            "eye_sha256": hashlib.sha256(eye.tobytes()).hexdigest(),
        }
        sched = {}
        for n0 in (1, 2, 3, 5, 16):
            for v in (1, 16, 100, 400):
                sched[f"{n0},{v}"] = [[int(k), int(c)] for k, c in
                                      get_candidates_and_visit_pairs(n0, v).items()]
        tables["halving"] = sched
        with open(os.path.join(GOLD, "tables.json"), "w") as f:
            json.dump(tables, f)

    # ---------------------------------------------------------------- board play-outs
    def snapshot(board):
        cells = np.array([board.board[p].value for p in board.onboard_pos], dtype=np.uint8)
        libs = np.array([board.strings.get_num_liberties(p) for p in board.onboard_pos],
                        dtype=np.int16)
        sizes = np.array([board.strings.string[board.strings.get_id(p)].get_size()
                          if board.strings.get_id(p) else 0 for p in board.onboard_pos],
                         dtype=np.int16)
        return cells, libs, sizes

    def candidates(board, color):
        c = board.get_all_legal_pos(color)
        c = [p for p in c if board.check_self_atari_stone(p, color) < 7
             and not board.is_complete_eye(p, color)]
        c.append(PASS)
        return c

    def padded(lst, n):
        out = np.full(n, -1, dtype=np.int16)
        out[:len(lst)] = lst
        return out

    def playout(seed, superko, max_plies, p_pass):
        rs = np.random.RandomState(seed)
        board = GoBoard(board_size=size, komi=7.0, check_superko=superko)
        rec = {k: [] for k in ("move", "color", "cells", "libs", "sizes", "ko_pos", "ko_move",
                               "pris", "legal_b", "legal_w", "cand_b", "cand_w", "score")}
        color = 1
        for _ in range(max_plies):
            legal = board.get_all_legal_pos(col(color))
            if not legal or rs.random_sample() < p_pass:
                mv = PASS
            else:
                mv = legal[rs.randint(len(legal))]
            board.put_stone(mv, col(color))
            cells, libs, sizes = snapshot(board)
            rec["move"].append(mv)
            rec["color"].append(color)
            rec["cells"].append(cells)
            rec["libs"].append(libs)
            rec["sizes"].append(sizes)
            rec["ko_pos"].append(board.ko_pos)
            rec["ko_move"].append(board.ko_move)
            rec["pris"].append(list(board.prisoner))
            rec["legal_b"].append(padded(board.get_all_legal_pos(Stone.BLACK), P + 1))
            rec["legal_w"].append(padded(board.get_all_legal_pos(Stone.WHITE), P + 1))
            rec["cand_b"].append(padded(candidates(board, Stone.BLACK), P + 1))
            rec["cand_w"].append(padded(candidates(board, Stone.WHITE), P + 1))
            rec["score"].append(board.count_score())
            color = 3 - color
        return board, {k: np.array(v) for k, v in rec.items()}

    n_games = 6 if size == 9 else 2
    plies = 150 if size == 9 else (240 if size == 13 else 420)
    board_fix = {}
    feat_boards = []
    for g in range(n_games):
        superko = (g % 2 == 0)
        _, rec = playout(1000 + g, superko, plies, 0.03)
        for k, v in rec.items():
            board_fix[f"g{g}_{k}"] = v
        board_fix[f"g{g}_superko"] = np.array(superko)
    np.savez_compressed(os.path.join(GOLD, f"board_{tag}.npz"), **board_fix)

    # ---------------------------------------------------------------- feature planes
    def replay(moves, colors, upto, superko=False):
        board = GoBoard(board_size=size, komi=7.0, check_superko=superko)
        for mv, c in zip(moves[:upto], colors[:upto]):
            board.put_stone(int(mv), col(int(c)))
        return board

    feat = {"game": [], "ply": [], "color": [], "planes": []}
    picks = [0, 1, 2, 3, 10, 25, 40, 60, 80, 100, 120, 140] if size == 9 else ([0, 1, 30, 120, 230] if size == 13 else [0, 1, 50, 200, 400])
    for g in range(min(n_games, 3)):
        moves = board_fix[f"g{g}_move"]
        colors = board_fix[f"g{g}_color"]
        for ply in picks:
            board = replay(moves, colors, ply)
            for c in (1, 2):
                feat["game"].append(g)
                feat["ply"].append(ply)
                feat["color"].append(c)
                feat["planes"].append(generate_input_planes(board, col(c), 0).astype(np.int8))
    # also a position right after a pass at move 1 and after two passes
    for k, seq in enumerate(([PASS], [PASS, PASS], [size + 3, PASS])):
        board = GoBoard(board_size=size)
        c = 1
        for mv in seq:
            board.put_stone(mv, col(c))
            c = 3 - c
        feat["game"].append(-(k + 1))
        feat["ply"].append(seq[0])
        feat["color"].append(c)
        feat["planes"].append(generate_input_planes(board, col(c), 0).astype(np.int8))
    # all eight symmetries (training-side featurisation, nn/feature.py:10 sym argument)
    sym_planes = []
    sym_meta = []
    for g in range(min(n_games, 2)):
        moves = board_fix[f"g{g}_move"]
        colors = board_fix[f"g{g}_color"]
        for ply in ([7, 33, 77] if size == 9 else [41]):
            board = replay(moves, colors, ply)
            for c in (1, 2):
                for sym in range(8):
                    sym_meta.append([g, ply, c, sym])
                    sym_planes.append(generate_input_planes(board, col(c), sym).astype(np.int8))
    feat_np = {k: np.array(v) for k, v in feat.items()}
    feat_np["sym_meta"] = np.array(sym_meta)
    feat_np["sym_planes"] = np.array(sym_planes)
    feat_np["special_seqs"] = np.array([[PASS, -9, -9], [PASS, PASS, -9], [size + 3, PASS, -9]])
    np.savez_compressed(os.path.join(GOLD, f"feat_{tag}.npz"), **feat_np)

    # ---------------------------------------------------------------- network outputs
    planes_all = feat_np["planes"].astype(np.float32)
    net_fix = {}
    for seed, gain in ((0, 1.0), (7, 1.5)):
        sd = make_state_dict(size, seed, gain)
        net = DualNet(torch.device("cpu"), board_size=size)
        missing = net.load_state_dict(sd)
        net.eval()
        nb = 16 if size == 9 else 4
        x = torch.from_numpy(planes_all[:nb])
        logits, vlogits = net.forward(x)
        pol, val = net.inference(x)
        lg2, val2 = net.inference_with_policy_logits(x)
        assert torch.equal(lg2, logits) and torch.equal(val, val2)
        net_fix[f"w{seed}_gain"] = np.array(gain)
        net_fix[f"w{seed}_planes"] = planes_all[:nb].astype(np.int8)
        net_fix[f"w{seed}_logits"] = logits.numpy()
        net_fix[f"w{seed}_vlogits"] = vlogits.numpy()
        net_fix[f"w{seed}_policy"] = pol.numpy()
        net_fix[f"w{seed}_value"] = val.numpy()
        # float64 re-computation for error budgeting
        net64 = DualNet(torch.device("cpu"), board_size=size).double()
        net64.load_state_dict({k: (v.double() if v.dtype == torch.float32 else v)
                               for k, v in sd.items()})
        net64.eval()
        l64, v64 = net64.forward(x.double())
        net_fix[f"w{seed}_logits64"] = l64.numpy()
        net_fix[f"w{seed}_vlogits64"] = v64.numpy()
        # B = 1 and B = 7 must give the same rows (checked here, not stored)
        p1, _ = net.inference(x[:1])
        assert np.allclose(p1.numpy(), pol.numpy()[:1], atol=1e-6)
    np.savez_compressed(os.path.join(GOLD, f"net_{tag}.npz"), **net_fix)

    # ---------------------------------------------------------------- RNG draws
    if size == 9:
        rng_fix = {}
        for seed in (0, 1, 12345):
            np.random.seed(seed)
            for n in (1, 2, 37, 82, 362):
                rng_fix[f"seed{seed}_dir{n}"] = np.random.dirichlet(alpha=np.ones(n))
            rng_fix[f"seed{seed}_gum82"] = np.random.gumbel(loc=0.0, scale=1.0, size=82)
            rng_fix[f"seed{seed}_dir5"] = np.random.dirichlet(alpha=np.ones(5))
            rng_fix[f"seed{seed}_uni"] = np.random.random_sample(4)
        np.savez_compressed(os.path.join(GOLD, "rng.npz"), **rng_fix)

    # ---------------------------------------------------------------- tree searches (StubNet)
    def tree_digest(tree):
        h = hashlib.sha256()
        for i in range(tree.num_nodes):
            nd = tree.node[i]
            n = nd.num_children
            h.update(np.array(nd.action[:n], dtype=np.int32).tobytes())
            h.update(nd.children_index[:n].astype(np.int32).tobytes())
            h.update(nd.children_visits[:n].astype(np.int32).tobytes())
            h.update(nd.children_virtual_loss[:n].astype(np.int32).tobytes())
            h.update(nd.children_value_sum[:n].astype(np.float64).tobytes())
            h.update(nd.children_policy[:n].astype(np.float64).tobytes())
            h.update(np.array([nd.node_visits, nd.virtual_loss], dtype=np.int64).tobytes())
        return h.hexdigest()

    def root_record(tree, mv):
        root = tree.get_root()
        n = root.num_children
        return {
            "move": int(mv), "num_nodes": int(tree.num_nodes), "n": int(n),
            "action": [int(a) for a in root.action[:n]],
            "child_visits": [int(v) for v in root.children_visits[:n]],
            "value_sum": [float(v).hex() for v in root.children_value_sum[:n]],
            "policy": [float(v).hex() for v in root.children_policy[:n]],
            "node_visits": int(root.node_visits),
            "node_value_sum": float(root.node_value_sum).hex(),
            "raw_value": float(root.raw_value).hex(),
            "digest": tree_digest(tree),
        }

    trees = []
    g0_moves = board_fix["g0_move"]
    g0_colors = board_fix["g0_color"]
    if size == 9:
        puct_cases = [
            # (seed, batch, visits, mode, cgos, start ply, superko)
            (0, 1, 100, "STRICT", False, 0, False),
            (0, 1, 100, "CONSTANT", False, 0, False),
            (1, 13, 100, "STRICT", False, 0, False),
            (1, 13, 100, "CONSTANT", True, 30, False),
            (2, 256, 1000, "STRICT", False, 0, False),
            (2, 256, 1000, "CONSTANT", False, 40, True),
            (3, 256, 1000, "STRICT", True, 90, True),
            (4, 64, 300, "STRICT", False, 130, True),
            (5, 8, 60, "STRICT", False, 147, False),
        ]
        gumbel_cases = [(1, 16, 0, True), (2, 100, 0, True), (3, 400, 0, True),
                        (4, 16, 50, True), (5, 100, 100, False), (6, 400, 120, True),
                        (7, 50, 148, True)]
    elif size == 13:
        puct_cases = [(0, 64, 200, "STRICT", False, 0, False),
                      (1, 32, 150, "CONSTANT", False, 100, True),
                      (2, 8, 60, "STRICT", True, 180, False)]
        gumbel_cases = [(1, 16, 0, True), (2, 100, 80, True)]
    else:
        puct_cases = [(0, 64, 200, "STRICT", False, 0, False),
                      (1, 64, 200, "STRICT", False, 200, True),
                      # round 6: BASELINE config[4] at full size (1 600 strict visits, NN batch 64) and a CONSTANT-mode
                      # search (early stop after every descent, mcts/time_manager.py) on a mid-game position
                      (2, 64, 1600, "STRICT", False, 0, False),
                      (3, 64, 400, "CONSTANT", False, 100, True)]
        gumbel_cases = [(1, 16, 0, True), (2, 100, 150, True)]

    for seed, batch, visits, mode, cgos, ply, superko in puct_cases:
        board = replay(g0_moves, g0_colors, ply, superko)
        color = 1 if ply == 0 else 3 - int(g0_colors[ply - 1])
        net = StubNet(salt=seed)
        tree = MCTSTree(net, tree_size=2048, batch_size=batch, cgos_mode=cgos)
        tm = TimeManager(TimeControl.STRICT_PLAYOUT if mode == "STRICT"
                         else TimeControl.CONSTANT_PLAYOUT, constant_visits=visits)
        np.random.seed(seed)
        stderr, sys.stderr = sys.stderr, open(os.devnull, "w")
        try:
            mv = tree.search_best_move(board, col(color), tm, {})
        finally:
            sys.stderr = stderr
        rec = root_record(tree, mv)
        rec["analysis_lz"] = tree.get_root().get_analysis(board, "lz", tree.get_pv_lists)
        rec["analysis_cgos"] = tree.get_root().get_analysis(board, "cgos", tree.get_pv_lists)
        rec.update(kind="puct", seed=seed, batch=batch, visits=visits, mode=mode, cgos=cgos,
                   ply=ply, superko=superko, color=color, batches=list(net.calls),
                   rng_after=float(np.random.random_sample()).hex())
        trees.append(rec)

    for seed, visits, ply, superko in gumbel_cases:
        board = replay(g0_moves, g0_colors, ply, superko)
        color = 1 if ply == 0 else 3 - int(g0_colors[ply - 1])
        net = StubNet(salt=100 + seed)
        tree = MCTSTree(net, tree_size=160 if visits <= 100 else 2048)
        tm = TimeManager(TimeControl.CONSTANT_PLAYOUT, constant_visits=visits)
        np.random.seed(seed)
        mv = tree.generate_move_with_sequential_halving(board, col(color), tm, True)
        rec = root_record(tree, mv)
        root = tree.get_root()
        rec.update(kind="gumbel", seed=seed, visits=visits, ply=ply, superko=superko,
                   color=color, batches=list(net.calls),
                   noise=[float(v).hex() for v in root.noise],
                   improved=[float(v).hex() for v in root.calculate_improved_policy()],
                   rng_after=float(np.random.random_sample()).hex())
        trees.append(rec)

    with open(os.path.join(GOLD, f"trees_{tag}.json"), "w") as f:
        json.dump(trees, f)

    # ---------------------------------------------------------------- one self-play game
    if size == 9:
        import selfplay.worker as worker_mod
        games = {}
        for k, visits in ((1, 16), (2, 16), (3, 50)):
            out = tempfile.mkdtemp(prefix="sgf_")
            worker_mod.load_network = lambda model_file_path, use_gpu, _k=k: StubNet(salt=200 + _k)
            random.seed(k)
            worker_mod.selfplay_worker(out, "/nonexistent/model.bin", [k], size, visits, False)
            games[f"{k},{visits}"] = open(os.path.join(out, f"{k}.sgf"), encoding="utf-8").read()
            shutil.rmtree(out, ignore_errors=True)
        with open(os.path.join(GOLD, "selfplay_games.json"), "w") as f:
            json.dump(games, f)
        # RL policy targets (nn/feature.py:60-102) for a few plies of game 1, all symmetries
        import re
        from nn.feature import generate_rl_target_data, generate_target_data
        sgf = games["1,16"]
        plies = re.findall(r";([BW])\[([a-t]{2})\]C\[([^\]]*)\]", sgf)
        board = GoBoard(board_size=size)
        targets = []
        for i, (c, mv, comment) in enumerate(plies[:40]):
            color = Stone.BLACK if c == "B" else Stone.WHITE
            pos = PASS if mv == "tt" else (ord(mv[0]) - 96) + (ord(mv[1]) - 96) * (size + 2)
            if i in (0, 5, 17, 39):
                for sym in range(8):
                    targets.append({"ply": i, "sym": sym, "comment": comment, "move": pos,
                                    "rl": [float(v).hex() for v in generate_rl_target_data(board, comment, sym)],
                                    "sl": [int(v) for v in generate_target_data(board, pos, sym)]})
            board.put_stone(pos, color)
        with open(os.path.join(GOLD, "rl_targets.json"), "w") as f:
            json.dump({"moves": [[c, mv] for c, mv, _ in plies[:40]], "targets": targets}, f)
    print("golden fixtures written for size", size)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=0)
    args = ap.parse_args()
    if args.size:
        worker(args.size)
    else:
        sys.exit(orchestrate())
