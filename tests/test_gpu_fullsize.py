"""BASELINE.json configs 3 and 4 at their full size on one GPU: Gumbel self-play with 16 and with
64 boards in lock-step at 400 simulations per move (per move and board 1 + 96 + 96 + 100 + 108
leaf evaluations), games to completion.

Parity: games 11..14 must be byte-identical to complete games recorded from the REFERENCE's
selfplay_worker at 400 visits (tests/golden/selfplay_games_400.json, tools/gen_golden_selfplay400.py);
every other game must come out identical whether it is played in the 16-board shard (one group),
the 64-board shard (auto grouping: pipelined groups with their own engine, stream and host thread)
or through the two-rank launcher - games are independent, lock-step batching must not leak
between boards."""
import json
import os
import random
import subprocess
import sys

import pytest

from tests.helpers import load_json

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VISITS = 400


def _flag(k):
    """never_resign as the reference worker draws it after random.seed(k) (worker.py:39,53)."""
    random.seed(k)
    random.choice([k])
    return random.randint(1, 10) == 1


def _read(d, i):
    return open(os.path.join(d, f"{i}.sgf"), encoding="utf-8").read()


@pytest.fixture(scope="module")
def shard16(tmp_path_factory):
    """cfg-3: 16 boards x 400 simulations, one lock-step group; games 11..26."""
    from oracle.stubnet import StubNet
    from tamago_amd.selfplay.worker import selfplay_shard
    out = str(tmp_path_factory.mktemp("cfg3"))
    idx = list(range(11, 27))
    stats = selfplay_shard(out, StubNet(salt=300), idx, 9, VISITS, boards=16,
                           never_resign_flags=[_flag(k) for k in idx], groups=1)
    return out, idx, stats


def test_cfg3_16_boards_400_sims_equal_reference_games(shard16):
    out, idx, stats = shard16
    golden = load_json("selfplay_games_400.json")
    assert stats["games"] == 16
    # every move costs its board 1 root evaluation + 400 simulations
    assert stats["leaf_evals"] == stats["moves"] * (VISITS + 1)
    for k in (11, 12, 13, 14):
        assert _read(out, k) == golden[f"{k},{VISITS}"], k


def test_cfg4_shard_64_boards_400_sims(shard16, tmp_path):
    """One cfg-4 shard (64 boards, default grouping = 2 pipelined groups): reference games and the
    16-board shard's games come out byte-identical."""
    from oracle.stubnet import StubNet
    from tamago_amd.selfplay.worker import selfplay_shard
    out16, idx16, _ = shard16
    golden = load_json("selfplay_games_400.json")
    idx = list(range(11, 75))
    stats = selfplay_shard(str(tmp_path), StubNet(salt=300), idx, 9, VISITS, boards=64,
                           never_resign_flags=[_flag(k) for k in idx])
    assert stats["games"] == 64 and stats["leaf_evals"] == stats["moves"] * (VISITS + 1)
    for k in (11, 12, 13, 14):
        assert _read(str(tmp_path), k) == golden[f"{k},{VISITS}"], k
    for k in idx16:
        assert _read(str(tmp_path), k) == _read(out16, k), k


def test_two_rank_launcher_on_one_gpu(tmp_path):
    """The cfg-4 launcher (python -m tamago_amd.selfplay) as two ranks under torch.distributed.run,
    both on cuda:0 (TG_SINGLE_DEVICE): disjoint, covering SGF sets in ONE record directory, aggregate
    statistics = sum of the shards, resume-by-skip."""
    env = dict(os.environ, TG_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29581",
               PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29581", "-m", "tamago_amd.selfplay",
           "--save-dir", str(tmp_path), "--num-data", "12", "--visits", "32", "--boards", "4",
           "--model", "/nonexistent/model.bin", "--never-resign", "--json"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    result = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    files = sorted(int(f[:-4]) for f in os.listdir(tmp_path / "1"))
    assert files == list(range(1, 13))
    assert result["shards"] == 2 and result["games"] == 12
    a, b = sorted(result["per_shard"], key=lambda s: s["rank"])
    assert (a["first"], a["last"], b["first"], b["last"]) == (1, 6, 7, 12)
    assert result["leaf_evals"] == a["leaf_evals"] + b["leaf_evals"] == result["moves"] * 33
    assert "games/hour" in out.stdout and "Failed to load /nonexistent/model.bin." in out.stdout
    # second run into the same directory: everything exists, nothing is played
    again = subprocess.run(cmd + ["--resume-dir", "1"], env=env, capture_output=True, text=True,
                           timeout=600, cwd=REPO)
    assert again.returncode == 0, again.stderr[-3000:]
    res2 = json.loads([ln for ln in again.stdout.splitlines() if ln.startswith("{")][-1])
    assert res2["games"] == 0


def test_eight_shards_share_one_gpu_without_collisions(tmp_path):
    """Multi-GPU readiness without the node: the stand-alone launcher with EIGHT shard processes on
    one GPU (TG_SINGLE_DEVICE) - no port, temp-dir, stream or record-file collisions; every shard
    pins itself to a private slice of the host cores."""
    env = dict(os.environ, TG_SINGLE_DEVICE="1",
               PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "tamago_amd.selfplay", "--save-dir", str(tmp_path), "--process", "8",
           "--num-data", "16", "--visits", "16", "--boards", "2", "--model", "/nonexistent/model.bin",
           "--never-resign", "--json"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    result = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert result["shards"] == 8 and result["games"] == 16
    assert sorted(int(f[:-4]) for f in os.listdir(tmp_path / "1")) == list(range(1, 17))
    assert [s["rank"] for s in result["per_shard"]] == list(range(8))
    assert all(s["games"] == 2 for s in result["per_shard"])


def test_bench_two_ranks_on_one_gpu():
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one rank per GPU), with both
    ranks on cuda:0 (TG_SINGLE_DEVICE) and gloo for the barrier / reductions (RCCL cannot put two ranks on one
    device): ONE JSON line from rank 0, whole-job aggregate over both ranks, weak scaling, max-over-ranks time;
    the cfg-4 leg (shrunk: 4 boards x 32 simulations) runs one self-play shard per rank."""
    env = dict(os.environ, TG_SINGLE_DEVICE="1", TG_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT="29583",
               PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29583", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "1", "--trees", "64", "--no-cpu-baseline", "--cfg4-boards", "4", "--cfg4-games", "6",
           "--cfg4-visits", "32"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 1 and res["scaling"] == "weak" and res["higher_is_better"] is True
    assert res["metric"].startswith("MCTS leaf-evals") and res["unit"] == "leaf-evals/s"
    # two ranks x 64 trees x 1001 leaf evaluations in the timed step
    assert abs(res["value"] * res["ms_per_step"] / 1e3 - 2 * 64 * 1001) < 1.0
    roof = res["roofline"]
    assert roof["bound"] == "mfma" and 0.0 < roof["frac"] < roof["mfma_issue_frac"] <= 1.0
    # frac = algorithmic FLOPs per launch / launch time / peak of the executed precision
    assert abs(roof["frac"] - roof["positions_per_launch"] * 2 * 36140823 / (roof["avg_launch_ms"] * 1e-3) / 1e12 / roof["peak"]) < 1e-9
    # BASELINE.json config[3] rides on the N > 1 launch: one Gumbel self-play shard per rank, aggregate on rank 0
    leg = res["cfg4_selfplay_shards"]
    assert leg["shards"] == 2 and [r["rank"] for r in leg["per_rank"]] == [0, 1]
    assert all(r["games"] == 6 and r["host_cores"] >= 1 for r in leg["per_rank"])
    assert abs(leg["value"] * leg["seconds"] - sum(r["moves"] for r in leg["per_rank"]) * 33) < 1.0


def test_bench_rank_failure_in_the_cfg4_leg_reaches_every_rank():
    """A rank whose cfg-4 shard raises (injected: TG_BENCH_FAIL_RANK=1) still takes part in the leg's collectives: the
    2-rank bench ends promptly - nobody waits in all_gather for the RCCL / gloo timeout -, rank 0 prints the headline line
    with the leg reported as an error naming the rank, exit code 0."""
    import time
    env = dict(os.environ, TG_SINGLE_DEVICE="1", TG_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT="29587",
               TG_BENCH_FAIL_RANK="1", PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29587", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "0", "--trees", "16", "--no-cpu-baseline", "--cfg4-boards", "4", "--cfg4-games", "4",
           "--cfg4-visits", "16"]
    t0 = time.time()
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=REPO)
    elapsed = time.time() - t0
    assert out.returncode == 0, out.stderr[-3000:]
    assert elapsed < 150, elapsed                       # (a hang would run into the 300 s / the collective's 10-minute timeout)
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["value"] > 0
    leg = res["cfg4_selfplay_shards"]
    assert leg["failed_ranks"] == [1] and "injected failure on rank 1" in leg["messages"]["1"], leg
