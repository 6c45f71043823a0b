#!/usr/bin/env python3
"""Full-size self-play goldens: complete games of the REFERENCE's selfplay_worker at the visit
budget of BASELINE.json configs 3/4 (400 simulations per move), evaluator = oracle.stubnet.StubNet(salt=300) for every game
(bit-reproducible outputs).  Build container only (imports /root/reference, never copies it);
writes tests/golden/selfplay_games_400.json = {"<index>,<visits>": SGF text} (data only).

    python tools/gen_golden_selfplay400.py            # games 11..14, one process each
"""
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden", "selfplay_games_400.json")
GAMES = (11, 12, 13, 14)
VISITS = 400


def one(k: int) -> str:
    """Reference game k (run with the reference on sys.path)."""
    import torch
    import selfplay.worker as worker_mod
    from oracle.stubnet import StubNet
    torch.set_grad_enabled(False)
    out = tempfile.mkdtemp(prefix="sgf400_")
    worker_mod.load_network = lambda model_file_path, use_gpu: StubNet(salt=300)
    random.seed(k)
    worker_mod.selfplay_worker(out, "/nonexistent/model.bin", [k], 9, VISITS, False)
    text = open(os.path.join(out, f"{k}.sgf"), encoding="utf-8").read()
    shutil.rmtree(out, ignore_errors=True)
    return text


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--game":
        print(json.dumps(one(int(sys.argv[2]))))
        sys.exit(0)
    if not os.path.isdir(REF):
        print("no reference checkout at", REF, "- nothing to do")
        sys.exit(0)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=REF + os.pathsep + REPO,
               OMP_NUM_THREADS="1")
    procs = [(k, subprocess.Popen([sys.executable, os.path.abspath(__file__), "--game", str(k)], env=env,
                                  cwd="/tmp", stdout=subprocess.PIPE, text=True)) for k in GAMES]
    games = {}
    for k, p in procs:
        out, _ = p.communicate()
        assert p.returncode == 0, k
        games[f"{k},{VISITS}"] = json.loads(out.strip().splitlines()[-1])
    with open(OUT, "w") as f:
        json.dump(games, f)
    print("wrote", OUT, {k: len(v) for k, v in games.items()})
