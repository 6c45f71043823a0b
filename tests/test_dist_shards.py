"""Multi-GPU path on CPU: self-play shards are independent (no data-path collective); the
only distributed step is the barrier + reduction bench.py uses for timing.  world_size 2,
gloo backend."""
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
