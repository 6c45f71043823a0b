#!/usr/bin/env python3
"""Golden GTP session: the REFERENCE's command loop (gtp/client.py) on a scripted session with the
deterministic stub network of the tree fixtures; stdout is recorded byte for byte.  Only works in the build container
(the reference is imported from /root/reference, never copied):

    PYTHONPATH=/root/reference:/root/repo python tools/gen_golden_gtp.py

-> tests/golden/gtp_session.json {script, seed, visits, batch_size, tree_size, stdout}, tests/golden/handicap.json."""
import io
import json
import os
import random
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = ("1 protocol_version\nname\nversion\nknown_command genmove\nknown_command foo\nboardsize 9\nclear_board\n"
          "komi 6.5\nget_komi\nplay b E5\nplay w C3\nplay b pass\n7 genmove w\n8 genmove b\nundo\nbogus\n"
          "play w A1\nplay w A1\ngenmove w\ntime_settings 60 0 0\ntime_left b 30 0\nfixed_handicap 2\nclear_board\n"
          "fixed_handicap 2\ngenmove w\nlz-genmove_analyze b 0\ncgos-genmove_analyze w\nquit\n")


def main():
    import gtp.client as ref_client
    from mcts.time_manager import TimeControl
    from oracle.stubnet import StubNet
    ref_client.load_network = lambda model_file_path, use_gpu: StubNet(8)
    visits, batch, tree = 60, 16, 256
    client = ref_client.GtpClient(9, True, "stub", False, False, False, 7.0, TimeControl.STRICT_PLAYOUT, visits, 5.0, 0.0,
                                  batch, tree, False, 0.0, 0.0)
    np.random.seed(5)
    random.seed(5)
    old_in, old_out = sys.stdin, sys.stdout
    sys.stdin, sys.stdout = io.StringIO(SCRIPT), io.StringIO()
    try:
        try:
            client.run()
        except (SystemExit, EOFError):
            pass
        out = sys.stdout.getvalue()
    finally:
        sys.stdin, sys.stdout = old_in, old_out
    path = os.path.join(REPO, "tests", "golden", "gtp_session.json")
    with open(path, "w") as f:
        json.dump({"script": SCRIPT, "seed": 5, "visits": visits, "batch_size": batch, "tree_size": tree, "stdout": out}, f, indent=1)
    print(out)
    print("wrote", path, len(out), "bytes")
    # the reference's handicap table (board/handicap.py), every (size, stones) it answers or refuses
    from board.handicap import get_handicap_coordinates
    table = {f"{size},{n}": get_handicap_coordinates(size, n) for size in range(5, 22) for n in range(0, 12)}
    with open(os.path.join(REPO, "tests", "golden", "handicap.json"), "w") as f:
        json.dump(table, f)


if __name__ == "__main__":
    main()
