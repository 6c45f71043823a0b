"""Self-play worker (W1 row of SURVEY.md section 8) on the GPU vs complete games recorded from
the reference's selfplay_worker (tests/golden/selfplay_games.json): byte-identical SGF text,
i.e. identical moves, improved-policy strings (.3e) and results."""
import os
import random

import numpy as np
import pytest

from tests.helpers import load_json

pytestmark = pytest.mark.gpu


def _flags(k):
    """never_resign as the reference worker draws it after random.seed(k) (worker.py:39,53)."""
    random.seed(k)
    random.choice([k])
    return random.randint(1, 10) == 1


@pytest.mark.parametrize("key", ["1,16", "2,16", "3,50"])
def test_reference_signature_worker_reproduces_reference_game(key, tmp_path, monkeypatch):
    from oracle.stubnet import StubNet
    import tamago_amd.nn.utility as util
    from tamago_amd.selfplay.worker import selfplay_worker
    k, visits = (int(v) for v in key.split(","))
    monkeypatch.setattr(util, "load_network", lambda **kw: StubNet(salt=200 + k))
    random.seed(k)
    selfplay_worker(str(tmp_path), "/nonexistent/model.bin", [k], 9, visits, True)
    got = open(os.path.join(tmp_path, f"{k}.sgf"), encoding="utf-8").read()
    assert got == load_json("selfplay_games.json")[key]


def test_lockstep_shard_equals_single_board_games(tmp_path):
    """Boards searched together (one forward pass per phase over all of them) play exactly
    the games they play alone; finished slots are refilled from the queue."""
    from oracle.stubnet import StubNet
    from tamago_amd.selfplay.worker import selfplay_shard
    golden = load_json("selfplay_games.json")
    solo = tmp_path / "solo"
    multi = tmp_path / "multi"
    solo.mkdir()
    multi.mkdir()
    idx = [1, 5, 6]
    flags = [_flags(1), True, False]
    for i, f in zip(idx, flags):
        selfplay_shard(str(solo), StubNet(salt=201), [i], 9, 16, boards=1, never_resign_flags=[f])
    stats = selfplay_shard(str(multi), StubNet(salt=201), idx, 9, 16, boards=2, never_resign_flags=flags)
    assert stats["games"] == 3 and stats["moves"] > 100
    for i in idx:
        a = open(solo / f"{i}.sgf", encoding="utf-8").read()
        b = open(multi / f"{i}.sgf", encoding="utf-8").read()
        assert a == b, i
    assert open(solo / "1.sgf", encoding="utf-8").read() == golden["1,16"]
    # two pipelined groups (own engine, HIP stream and host thread each) play the same games
    piped = tmp_path / "piped"
    piped.mkdir()
    stats = selfplay_shard(str(piped), StubNet(salt=201), idx, 9, 16, boards=3, never_resign_flags=flags,
                           groups=2)
    assert stats["games"] == 3
    for i in idx:
        assert open(solo / f"{i}.sgf", encoding="utf-8").read() == open(piped / f"{i}.sgf", encoding="utf-8").read(), i
    # resume-by-skip (worker.py:47-48): nothing left to do
    again = selfplay_shard(str(multi), StubNet(salt=201), idx, 9, 16, boards=2, never_resign_flags=flags)
    assert again["games"] == 0


def test_one_call_per_move_path_equals_the_phase_by_phase_path(tmp_path):
    """With a DualNet the shard plays each move through ONE library call (tg_selfplay_play_move: root
    evaluation, noise, schedule, every phase with its forward pass, records, play).  Wrapped so that it is
    not recognised as a DualNet, the same network goes through the phase-by-phase path (host evaluator API).
    Same forward kernel, same bits: the games must be byte-identical."""
    import torch
    from tamago_amd.nn.network.dual_net import DualNet
    from tamago_amd.selfplay.worker import selfplay_shard

    class HostOnly:
        def __init__(self, net):
            self.net = net

        def inference(self, planes):
            return self.net.inference(planes)

        def inference_with_policy_logits(self, planes):
            return self.net.inference_with_policy_logits(planes)

    torch.manual_seed(21)
    net = DualNet(torch.device("cuda:0"), 9)
    idx = list(range(1, 11))
    flags = [i % 3 == 0 for i in idx]
    fast, slow = tmp_path / "fast", tmp_path / "slow"
    fast.mkdir(), slow.mkdir()
    a = selfplay_shard(str(fast), net, idx, 9, 48, boards=4, never_resign_flags=flags)
    b = selfplay_shard(str(slow), HostOnly(net), idx, 9, 48, boards=4, never_resign_flags=flags)
    assert a == b and a["games"] == 10
    for i in idx:
        assert open(fast / f"{i}.sgf").read() == open(slow / f"{i}.sgf").read(), i
    # ONE board at a time: every game's end leaves the group with nothing but a freshly started game - a call that launches no
    # phase at all.  (Round 6: the random window of such a call was installed behind the note of each board's cursor, and the next
    # move read a wrong root width off it - "82 root children but its root expansion consumed 81 draws"; it takes all games of a
    # group ending on the same move, which larger groups rarely do.)  Same games again, also from two boards in two lanes.
    for name, boards, lanes in (("solo", 1, 1), ("two_lanes_of_one", 2, 2)):
        d = tmp_path / name
        d.mkdir()
        c = selfplay_shard(str(d), net, idx, 9, 48, boards=boards, never_resign_flags=flags, groups=1, lanes=lanes)
        assert c == a, name
        for i in idx:
            assert open(d / f"{i}.sgf").read() == open(fast / f"{i}.sgf").read(), (name, i)


def test_gumbel_kernel_variants_write_the_same_games():
    """select_gumbel_pipe_kernel with two, four, six and ten workers per tree (ten is the default up to 28 trees, six up to
    128; every wave walks its share of a phase's root children, a worker takes whole entries: expansion, the step into the new
    node, the leaf and its repeats), the same kernel with every entry sent one by one through its job ring - what a phase with
    a path longer than the per-entry buffers does (TG_GUMBEL_ONE_BY_ONE=1) - and the one-wavefront kernel
    (TG_SELECT_SERIAL=1) play byte-identical games: 8 lock-step boards, 64 simulations per move."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_gumbel_games.py")
    outs = []
    for variant in ("serial", "2", "4", "6", "10", "one-by-one", "one-by-one 2"):
        env = dict(os.environ)
        for k in ("TG_SELECT_SERIAL", "TG_GUMBEL_WORKERS", "TG_GUMBEL_ONE_BY_ONE"):
            env.pop(k, None)
        if variant == "serial":
            env["TG_SELECT_SERIAL"] = "1"
        elif variant.startswith("one-by-one"):
            env["TG_GUMBEL_ONE_BY_ONE"] = "1"
            if variant.endswith(" 2"):
                env["TG_GUMBEL_WORKERS"] = "2"
        else:
            env["TG_GUMBEL_WORKERS"] = variant
        res = subprocess.run([sys.executable, script, "8", "12", "64"], env=env, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, (variant, res.stderr[-2000:])
        outs.append(res.stdout.strip().splitlines()[-1])
    assert len(set(outs)) == 1, outs


def test_gumbel_kernels_agree_at_800_simulations():
    """More than 512 leaf slots per tree (800 simulations per move) with phases of ~200 descents: the pipelined kernel is chosen
    by the phases' sizes, not by the slot count (until the end of round 6 such a shard ran on the one-wavefront kernel), and
    plays the one-wavefront kernel's games."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_gumbel_games.py")
    outs = []
    for variant in ("serial", "default"):
        env = dict(os.environ)
        for k in ("TG_SELECT_SERIAL", "TG_GUMBEL_WORKERS", "TG_GUMBEL_ONE_BY_ONE"):
            env.pop(k, None)
        if variant == "serial":
            env["TG_SELECT_SERIAL"] = "1"
        res = subprocess.run([sys.executable, script, "3", "3", "800"], env=env, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, (variant, res.stderr[-2000:])
        outs.append(res.stdout.strip().splitlines()[-1])
    assert len(set(outs)) == 1, outs


@pytest.mark.parametrize("size,workers", [(13, ("6", "2")), (19, ("4", "2"))])
def test_gumbel_kernel_variants_write_the_same_games_on_larger_boards(size, workers):
    """BOARD_SIZE = 13 / 19: select_gumbel_pipe_kernel with its two worker counts per tree (six / four are the defaults up to
    128 trees), the same kernel one by one through its job ring, and the one-wavefront kernel play byte-identical games: 4
    lock-step boards, 24 simulations per move."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_gumbel_games.py")
    outs = []
    for variant in ("serial",) + workers + ("one-by-one",):
        env = dict(os.environ)
        for k in ("TG_SELECT_SERIAL", "TG_GUMBEL_WORKERS", "TG_GUMBEL_ONE_BY_ONE"):
            env.pop(k, None)
        if variant == "serial":
            env["TG_SELECT_SERIAL"] = "1"
        elif variant == "one-by-one":
            env["TG_GUMBEL_ONE_BY_ONE"] = "1"
        else:
            env["TG_GUMBEL_WORKERS"] = variant
        res = subprocess.run([sys.executable, script, "4", "4", "24", str(size)], env=env, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, (variant, res.stderr[-2000:])
        outs.append(res.stdout.strip().splitlines()[-1])
    assert len(set(outs)) == 1, outs


def test_13x13_selfplay_one_call_path_equals_the_phase_by_phase_path(tmp_path):
    """BOARD_SIZE = 13 (board/constant.py:4; built in round 6: board-size-generic tree kernels, the exact-fp32 forward kernel):
    a shard's games through the one-call chained path and through the phase-by-phase host path are the same files, and a game
    played alone equals its copy in the shard."""
    import torch
    from tamago_amd.nn.network.dual_net import DualNet
    from tamago_amd.selfplay.worker import selfplay_shard

    class HostOnly:
        def __init__(self, net):
            self.net = net

        def inference(self, planes):
            return self.net.inference(planes)

        def inference_with_policy_logits(self, planes):
            return self.net.inference_with_policy_logits(planes)

    torch.manual_seed(22)
    net = DualNet(torch.device("cuda:0"), 13)
    idx = list(range(1, 6))
    flags = [i % 2 == 0 for i in idx]
    fast, slow, solo = tmp_path / "fast", tmp_path / "slow", tmp_path / "solo"
    fast.mkdir(), slow.mkdir(), solo.mkdir()
    a = selfplay_shard(str(fast), net, idx, 13, 24, boards=3, never_resign_flags=flags)
    b = selfplay_shard(str(slow), HostOnly(net), idx, 13, 24, boards=3, never_resign_flags=flags)
    assert a == b and a["games"] == 5
    for i in idx:
        text = open(fast / f"{i}.sgf").read()
        assert text == open(slow / f"{i}.sgf").read(), i
        assert "SZ[13]" in text
    selfplay_shard(str(solo), net, idx[:1], 13, 24, boards=1, never_resign_flags=flags[:1])
    assert open(solo / "1.sgf").read() == open(fast / "1.sgf").read()
