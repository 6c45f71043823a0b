"""Multi-GPU path on CPU: self-play shards are independent (no data-path collective); the
only distributed step is the barrier + reduction bench.py uses for timing.  world sizes 2
and 8 (the reference's selfplay_main.py:44-65 runs 8 worker shards), gloo backend."""
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["TG_REPO"])
import torch, torch.distributed as dist
from tamago_amd.selfplay.worker import shard_indices
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
games = list(range(100, 111))
mine = shard_indices(games, world, rank)
# pretend every game yields 1001 leaf evaluations; aggregate like bench.py does
leaves = torch.tensor([1001.0 * len(mine)], dtype=torch.float64)
elapsed = torch.tensor([1.0 + rank], dtype=torch.float64)
dist.barrier()
dist.all_reduce(leaves, op=dist.ReduceOp.SUM)
dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
gathered = [None] * world
dist.all_gather_object(gathered, mine)
if rank == 0:
    flat = [g for part in gathered for g in part]
    assert flat == games, flat                       # disjoint, covering, order-preserving
    assert abs(len(gathered[0]) - len(gathered[1])) <= 1
    assert leaves.item() == 1001.0 * len(games) and elapsed.item() == 2.0
    print("OK", leaves.item() / elapsed.item())
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_shards_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, TG_REPO=REPO, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
                          "29571", str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "OK" in out.stdout


def test_shard_indices_properties():
    from tamago_amd.selfplay.worker import shard_indices
    for n in (0, 1, 7, 8, 64, 513):
        for world in (1, 2, 3, 8):
            parts = [shard_indices(list(range(n)), world, r) for r in range(world)]
            assert [g for p in parts for g in p] == list(range(n))
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1


LAUNCHER = r'''
import json, os, sys
sys.path.insert(0, os.environ["TG_REPO"])
from tamago_amd.selfplay import main as launcher
from tamago_amd.selfplay.worker import shard_indices

def fake_shard(args, rank, world, local_rank, record_dir):
    # what run_shard reports, without a GPU: this rank's block of game indices
    assert os.path.isdir(record_dir)
    mine = shard_indices(list(range(1, args.num_data + 1)), world, rank)
    open(os.path.join(record_dir, f"rank{rank}.txt"), "w").write(" ".join(map(str, mine)))
    return {"games": len(mine), "moves": 10 * len(mine), "leaf_evals": 401 * 10 * len(mine), "seconds": 0.5 + rank,
            "rank": rank, "device": local_rank, "host_cores": launcher.pin_host_threads(local_rank, world),
            "first": mine[0], "last": mine[-1]}

launcher.run_shard = fake_shard
result = launcher.main(["--save-dir", os.environ["TG_SAVE"], "--num-data", "11", "--visits", "400", "--boards", "64",
                        "--size", "9", "--json"])
if int(os.environ["RANK"]) == 0:
    assert result["shards"] == 2 and result["games"] == 11 and result["leaf_evals"] == 401 * 110
    assert result["seconds"] >= 1.5                      # slowest rank
    assert sorted(s["rank"] for s in result["per_shard"]) == [0, 1]
    print("LAUNCH-OK")
'''


def test_selfplay_launcher_two_ranks_gloo(tmp_path):
    """The config[3] launcher under torch.distributed.run, world size 2, gloo: record directory chosen by rank 0
    and broadcast, disjoint game blocks, aggregate over ranks, slowest rank's clock (the shard itself is stubbed -
    the GPU run of the same launch is tests/test_gpu_fullsize.py)."""
    script = tmp_path / "launch.py"
    script.write_text(LAUNCHER)
    save = tmp_path / "archive"
    save.mkdir()
    (save / "3").mkdir()                                   # existing record directories: next is 4
    env = dict(os.environ, TG_REPO=REPO, TG_SAVE=str(save), MASTER_ADDR="127.0.0.1", MASTER_PORT="29573")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29573", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "LAUNCH-OK" in out.stdout
    r0 = (save / "4" / "rank0.txt").read_text().split()
    r1 = (save / "4" / "rank1.txt").read_text().split()
    assert [int(g) for g in r0 + r1] == list(range(1, 12))


FAILING = r'''
import os, sys
sys.path.insert(0, os.environ["TG_REPO"])
from tamago_amd.selfplay import main as launcher

def fake_shard(args, rank, world, local_rank, record_dir):
    if rank == 1:
        raise RuntimeError("GPU fell off the bus")
    return {"games": 3, "moves": 30, "leaf_evals": 401 * 30, "seconds": 0.25, "rank": rank, "device": local_rank,
            "host_cores": 1, "first": 1, "last": 3}

launcher.run_shard = fake_shard
launcher.main(["--save-dir", os.environ["TG_SAVE"], "--num-data", "6", "--visits", "400", "--boards", "4", "--json"])
'''


def test_launcher_rank_failure_reaches_every_rank(tmp_path):
    """A shard that raises must not strand the other ranks in a collective: its error travels through
    the all-gather, the surviving shards are still reported, every rank exits non-zero."""
    script = tmp_path / "failing.py"
    script.write_text(FAILING)
    save = tmp_path / "archive"
    save.mkdir()
    env = dict(os.environ, TG_REPO=REPO, TG_SAVE=str(save), MASTER_ADDR="127.0.0.1", MASTER_PORT="29575")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29575", str(script)],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert "GPU fell off the bus" in out.stderr + out.stdout
    assert "shard 0" in out.stdout and "games/hour" in out.stdout          # the surviving shard's report


def _torchrun(nproc, port, script_args, env, timeout=600):
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
                           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args,
                          env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_rank_plumbing_at_world_size_8(tmp_path):
    """bench.py's N-rank code path at N = 8, before an 8-GPU node shows up: TG_BENCH_DRY_RUN=1 keeps the collectives of the
    GPU path in their order (barrier / timed steps / barrier, MAX of the clock, SUM of the counts, the host-cost gather, the
    cfg-4 gather) and swaps the workload for a stub.  Rank 0 prints ONE line: the aggregate is the sum over eight ranks over
    the slowest rank's clock; a failure injected into a MIDDLE rank's cfg-4 shard reaches rank 0 as an error row instead of
    stranding seven ranks in a collective."""
    import json
    env = dict(os.environ, TG_BENCH_DRY_RUN="1", MASTER_ADDR="127.0.0.1")
    env.pop("TG_BENCH_FAIL_RANK", None)
    bench = os.path.join(REPO, "bench.py")
    out = _torchrun(8, 29581, [bench, "--gpus", "8", "--steps", "3", "--warmup", "1", "--trees", "64", "--cfg4-games", "4"], env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["steps"] == 3 and line["scaling"] == "weak" and "DRY RUN" in line["data"]
    per_step = 64 * 1001
    # whole-job aggregate: eight ranks' units over the max-over-ranks time (every stub step sleeps 10 - 30 ms)
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - 8 * per_step) < 1e-6 * 8 * per_step
    assert line["ms_per_step"] >= 30.0
    assert [r["rank"] for r in line["host"]["per_rank"]] == list(range(8))
    leg = line["cfg4_selfplay_shards"]
    assert leg["shards"] == 8 and [r["rank"] for r in leg["per_rank"]] == list(range(8))
    assert leg["seconds"] >= 0.15 and "host_cpu_s_per_1e6_leaf_evals" in leg
    assert abs(leg["value"] * leg["seconds"] - 8 * 401 * 60 * 4) < 1e-6 * 8 * 401 * 60 * 4
    # a middle rank's shard fails
    env["TG_BENCH_FAIL_RANK"] = "3"
    out = _torchrun(8, 29583, [bench, "--gpus", "8", "--steps", "1", "--warmup", "0", "--trees", "8", "--cfg4-games", "2"], env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    leg = line["cfg4_selfplay_shards"]
    assert leg["failed_ranks"] == [3] and "injected failure on rank 3" in leg["messages"]["3"]
    assert line["value"] > 0                                 # the headline of the same run stands


LAUNCHER8 = r'''
import json, os, sys
sys.path.insert(0, os.environ["TG_REPO"])
from tamago_amd.selfplay import main as launcher
from tamago_amd.selfplay.worker import shard_indices

def fake_shard(args, rank, world, local_rank, record_dir):
    assert os.path.isdir(record_dir) and world == 8
    if os.environ.get("TG_FAIL_RANK") == str(rank):
        raise RuntimeError(f"shard {rank}: GPU fell off the bus")
    mine = shard_indices(list(range(1, args.num_data + 1)), world, rank)
    open(os.path.join(record_dir, f"rank{rank}.txt"), "w").write(" ".join(map(str, mine)))
    return {"games": len(mine), "moves": 10 * len(mine), "leaf_evals": 401 * 10 * len(mine), "seconds": 0.1 * (1 + rank % 4),
            "rank": rank, "device": local_rank, "host_cores": launcher.pin_host_threads(local_rank, world),
            "first": mine[0], "last": mine[-1]}

launcher.run_shard = fake_shard
result = launcher.main(["--save-dir", os.environ["TG_SAVE"], "--num-data", "67", "--visits", "400", "--boards", "64",
                        "--size", "9", "--json"])
if int(os.environ["RANK"]) == 0:
    assert result["shards"] == 8 and result["games"] == 67 and result["leaf_evals"] == 401 * 670
    assert result["seconds"] >= 0.4                      # slowest rank
    assert sorted(s["rank"] for s in result["per_shard"]) == list(range(8))
    print("LAUNCH8-OK")
'''


def test_selfplay_launcher_eight_ranks_gloo(tmp_path):
    """BASELINE.json config[3]'s launcher at its real world size (selfplay_main.py:44-65: 8 worker shards): one record
    directory chosen by rank 0, eight disjoint contiguous game blocks covering 1 .. 67 in rank order, the aggregate over
    eight ranks on the slowest rank's clock; then the same launch with a MIDDLE rank (5) raising: every rank exits
    non-zero, the error text and the surviving shards' report reach the output."""
    script = tmp_path / "launch8.py"
    script.write_text(LAUNCHER8)
    save = tmp_path / "archive"
    save.mkdir()
    env = dict(os.environ, TG_REPO=REPO, TG_SAVE=str(save), MASTER_ADDR="127.0.0.1")
    env.pop("TG_FAIL_RANK", None)
    out = _torchrun(8, 29585, [str(script)], env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "LAUNCH8-OK" in out.stdout
    games = []
    for r in range(8):
        mine = [int(g) for g in (save / "1" / f"rank{r}.txt").read_text().split()]
        assert mine == list(range(mine[0], mine[0] + len(mine))) and 8 <= len(mine) <= 9
        games += mine
    assert games == list(range(1, 68))
    env["TG_FAIL_RANK"] = "5"
    out = _torchrun(8, 29587, [str(script)], env, timeout=180)
    assert out.returncode != 0
    assert "shard 5: GPU fell off the bus" in out.stderr + out.stdout
    assert "shard 0" in out.stdout and "shard 7" in out.stdout and "games/hour" in out.stdout


def test_numa_aware_pinning_slices_the_gpus_node():
    """pin_host_threads takes a rank's core slice from the cores of the NUMA node its GPU hangs off (when the platform
    says which) and falls back to contiguous slices of all visible cores otherwise."""
    import os
    from tamago_amd.selfplay.main import parse_cpulist, pin_host_threads
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert parse_cpulist("") == []
    saved = os.sched_getaffinity(0)
    saved_env = os.environ.get("TG_HOST_THREADS")
    cores = sorted(saved)
    try:
        if len(cores) >= 4:
            node = cores[len(cores) // 2:]                     # pretend the GPU's node is the upper half of the cores
            os.environ["TG_SINGLE_DEVICE"] = "1"               # (both ranks of the test "share" that node)
            n = pin_host_threads(1, 2, device_index=None, numa_cores=node)
            mine = sorted(os.sched_getaffinity(0))
            assert n == len(mine) == len(node) // 2 and set(mine) <= set(node)
            assert mine == node[len(node) // 2:][:n]           # rank 1 of 2 on that node: its second half
            os.sched_setaffinity(0, saved)
        # no NUMA information: contiguous slices of everything visible
        n = pin_host_threads(0, 2)
        assert sorted(os.sched_getaffinity(0)) == cores[:max(1, len(cores) // 2)] and n == max(1, len(cores) // 2)
    finally:
        os.sched_setaffinity(0, saved)
        os.environ.pop("TG_SINGLE_DEVICE", None)
        if saved_env is None:
            os.environ.pop("TG_HOST_THREADS", None)
        else:
            os.environ["TG_HOST_THREADS"] = saved_env


def test_cpu_selfplay_worker_counts_leaf_evaluations():
    """oracle/cpu_selfplay.py (bench.py's cpu_selfplay leg): a 16-simulation Gumbel move is 1 + 16 leaf evaluations."""
    from oracle.cpu_selfplay import run
    out = run(seconds=1.5, visits=16, seed=3, threads=1)
    assert out["moves"] >= 1 and out["leaf_evals"] == out["moves"] * 17, out
