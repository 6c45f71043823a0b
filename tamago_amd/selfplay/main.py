"""Self-play launcher for one MI355X node: the reference's ``selfplay_main.py:16-72`` with one
worker shard per GPU instead of N processes on ``cuda:0``.

    # 8 shards x 64 boards (BASELINE.json config 4), one process per GPU, no RCCL:
    python -m tamago_amd.selfplay --save-dir archive --process 8 --boards 64 --num-data 10000 --visits 400
    # or under torchrun (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE from the environment):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m tamago_amd.selfplay --save-dir archive --num-data 10000 --visits 400

The game indices 1..num_data are split into contiguous slices (selfplay_main.py:44-47 ->
``shard_indices``), rank r drives its slice on device r with ``--boards`` games in lock-step
(``selfplay_shard``), records land in ``<save-dir>/<n>/<index>.sgf`` (n = highest existing sub-
directory + 1, selfplay_main.py:48-54; ``--resume-dir n`` continues directory n and skips the files
that exist, worker.py:47-48).  Shards never communicate: the only cross-rank steps are agreeing on
the directory before the start and summing the statistics at the end, both over the launcher's own
pipes (spawn mode) or a gloo group (torchrun mode) - host side, no device collective.
"""
import argparse
import glob
import json
import math
import os
import sys
import time

SELF_PLAY_VISITS = 16            # learning_param.py:40
NUM_SELF_PLAY_GAMES = 10000      # learning_param.py:46


def parse(argv=None):
    ap = argparse.ArgumentParser(prog="python -m tamago_amd.selfplay", description=__doc__.split("\n\n")[0])
    ap.add_argument("--save-dir", default="archive")
    ap.add_argument("--process", type=int, default=0,
                    help="worker shards = GPUs used (default: WORLD_SIZE under torchrun, else every visible GPU)")
    ap.add_argument("--num-data", type=int, default=NUM_SELF_PLAY_GAMES)
    ap.add_argument("--size", type=int, default=9)
    ap.add_argument("--use-gpu", type=lambda v: str(v).lower() in ("1", "true", "yes"), default=True)
    ap.add_argument("--visits", type=int, default=SELF_PLAY_VISITS)
    ap.add_argument("--model", default=os.path.join("model", "rl-model.bin"))
    ap.add_argument("--boards", type=int, default=64, help="games a shard advances in lock-step")
    ap.add_argument("--groups", type=int, default=0, help="pipelined lock-step groups per shard (0 = auto)")
    ap.add_argument("--resume-dir", type=int, default=0, help="continue <save-dir>/<n> instead of opening a new one")
    ap.add_argument("--never-resign", action="store_true", help="play every game to the end (benchmarks)")
    ap.add_argument("--json", action="store_true", help="print the aggregate as one JSON line as well")
    return ap.parse_args(argv)


def next_record_dir(save_dir: str) -> int:
    """selfplay_main.py:48-52: highest numeric sub-directory + 1."""
    found = [0]
    for path in glob.glob(os.path.join(save_dir, "*")):
        name = os.path.split(path)[-1]
        if name.isdigit():
            found.append(int(name))
    return max(found) + 1


def gpu_numa_cores(device_index: int):
    """Cores of the NUMA node the GPU hangs off (/sys/class/drm/card*/device/numa_node + .../local_cpulist), or None
    when the platform does not say (no sysfs entry, numa_node -1, single-node host)."""
    try:
        import torch
        bus = torch.cuda.get_device_properties(device_index).pci_bus_id
        dom = torch.cuda.get_device_properties(device_index).pci_domain_id
        dev = torch.cuda.get_device_properties(device_index).pci_device_id
        path = f"/sys/bus/pci/devices/{dom:04x}:{bus:02x}:{dev:02x}.0"
        with open(os.path.join(path, "numa_node")) as f:
            if int(f.read().strip()) < 0:
                return None
        with open(os.path.join(path, "local_cpulist")) as f:
            return parse_cpulist(f.read())
    except Exception:
        return None


def parse_cpulist(text: str):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def pin_host_threads(local_rank: int, local_world: int, device_index=None, numa_cores=None) -> int:
    """Give this shard a private slice of the host's cores: the group threads and the library's random-stream
    generator threads of 8 shards must not pile onto the same cores (host-CPU contention is the
    only scaling loss).  The slice is taken from the cores of the NUMA node the rank's GPU hangs off when the
    platform says which (the ranks whose GPUs share a node split that node's cores among themselves by their
    position in local-rank order); otherwise contiguous slices of all visible cores.  Returns the slice size."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1
    mine = None
    if numa_cores is None and device_index is not None:
        numa_cores = gpu_numa_cores(device_index)
    if numa_cores:
        local = [c for c in cores if c in set(numa_cores)]
        # the other local ranks are assumed to map onto the devices the way this one does (rank r -> device r mod n):
        # ranks on the same node = those whose device reports the same core list
        peers = sharing = 0
        all_known = True                                 # NUMA slices or contiguous slices: the same rule for every local rank,
        try:                                             # or a rank without NUMA information lands on another's slice
            import torch
            n_dev = max(1, torch.cuda.device_count())
            for r in range(local_world):
                theirs = numa_cores if os.environ.get("TG_SINGLE_DEVICE") else gpu_numa_cores(r % n_dev)
                if theirs is None:
                    all_known = False
                if theirs == numa_cores:
                    if r < local_rank:
                        peers += 1
                    sharing += 1
        except Exception:
            peers, sharing = local_rank, local_world
        if local and sharing and all_known:
            per = max(1, len(local) // sharing)
            mine = local[peers * per:(peers + 1) * per]
    if not mine:
        per = max(1, len(cores) // max(1, local_world))
        mine = cores[local_rank * per:(local_rank + 1) * per] or cores
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return len(cores)
    os.environ["TG_HOST_THREADS"] = str(max(1, min(16, len(mine) // 2)))
    return len(mine)


def run_shard(args, rank: int, world: int, local_rank: int, record_dir: str) -> dict:
    """One worker shard: this process drives ONE GPU."""
    import torch
    from tamago_amd.nn.utility import load_network
    from tamago_amd.selfplay.worker import selfplay_shard, shard_indices
    n_dev = max(1, torch.cuda.device_count())
    # more worker processes than GPUs (the reference's --process N puts N workers on ONE device, nn/utility.py:22):
    # the shards share the devices round-robin
    device_index = 0 if os.environ.get("TG_SINGLE_DEVICE") else local_rank % n_dev
    # ranks of THIS node against the GPUs of this node (a 16-rank job on two 8-GPU nodes shares nothing)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    shared = bool(os.environ.get("TG_SINGLE_DEVICE")) or local_world > n_dev
    if shared:
        # several shards share a device: the three-workgroups-per-tree selection kernel needs all of a tree's workgroups
        # resident at once, which another shard's kernels can prevent - keep to the one-workgroup kernels
        # (TG_SHARED_DEVICE is the product switch for both this and the 19x19 pair kernel; read when the handles are created)
        os.environ.setdefault("TG_SHARED_DEVICE", "1")
    cores = pin_host_threads(local_rank, local_world, device_index)
    torch.cuda.set_device(device_index)
    network = load_network(model_file_path=args.model, use_gpu=args.use_gpu, board_size=args.size,
                           device_index=device_index)
    if shared and hasattr(network, "set_shared_device"):
        network.set_shared_device(True)                    # ... and so does the banded 19x19 forward (net_forward_band.hip)
    mine = shard_indices(list(range(1, args.num_data + 1)), world, rank)
    flags = [True] * len(mine) if args.never_resign else None
    t0 = time.perf_counter()
    stats = selfplay_shard(record_dir, network, mine, args.size, args.visits, boards=args.boards,
                           device_index=device_index, never_resign_flags=flags, groups=args.groups)
    torch.cuda.synchronize()
    stats = dict(stats)
    stats.update(seconds=time.perf_counter() - t0, rank=rank, device=device_index, host_cores=cores,
                 first=mine[0] if mine else 0, last=mine[-1] if mine else 0)
    return stats


def aggregate(per_rank, elapsed: float, visits: int, boards: int) -> dict:
    games = sum(s["games"] for s in per_rank)
    leaves = sum(s["leaf_evals"] for s in per_rank)
    return {"shards": len(per_rank), "boards_per_shard": boards, "visits": visits, "games": games,
            "moves": sum(s["moves"] for s in per_rank), "leaf_evals": leaves, "seconds": elapsed,
            "games_per_hour": 3600.0 * games / elapsed if elapsed > 0 else 0.0,
            "leaf_evals_per_s": leaves / elapsed if elapsed > 0 else 0.0,
            "per_shard": per_rank}


def report(result: dict, as_json: bool):
    for s in result["per_shard"]:
        print(f"shard {s['rank']} (cuda:{s['device']}, {s['host_cores']} host cores): games {s['first']}..{s['last']}: "
              f"{s['games']} played, {s['leaf_evals'] / max(s['seconds'], 1e-9):.0f} leaf-evals/s")
    # selfplay_main.py:70-72
    print(f"{result['seconds']:3f} seconds, {result['games_per_hour']:3f} games/hour")
    print(f"{result['leaf_evals_per_s']:.0f} leaf-evals/s over {result['shards']} shard(s) x "
          f"{result['boards_per_shard']} boards")
    if as_json:
        print(json.dumps(result))


def _spawn_entry(argv, rank, world, record_dir, conn):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world))
    try:
        conn.send(run_shard(parse(argv), rank, world, rank, record_dir))
    except BaseException as exc:            # surfaced by the parent
        conn.send({"error": repr(exc), "rank": rank})
        raise
    finally:
        conn.close()


def main(argv=None) -> dict:
    args = parse(argv)
    if args.visits < 2 or args.num_data < 1:
        raise SystemExit("--visits must be >= 2 and --num-data >= 1")
    under_torchrun = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if under_torchrun:
        # one rank per GPU, started by torch.distributed.run; gloo carries two host-side objects
        import torch.distributed as dist
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        local_rank = int(os.environ.get("LOCAL_RANK", rank))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        # Every rank reaches every collective whatever happens to it: a failure (record directory, shard) travels as an
        # {"error": ...} object, so that no rank is left waiting in a gloo collective and partial results are reported.
        box = [None]
        if rank == 0:
            try:
                n = args.resume_dir or next_record_dir(args.save_dir)
                os.makedirs(os.path.join(args.save_dir, str(n)), exist_ok=True)
                box = [n]
            except Exception as exc:
                box = [{"error": f"record directory: {exc!r}"}]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        if isinstance(box[0], dict):
            if world > 1:
                dist.destroy_process_group()
            raise SystemExit(f"self-play launcher: {box[0]['error']}")
        if rank == 0:
            print(f"Self play visits : {args.visits}")
        record_dir = os.path.join(args.save_dir, str(box[0]))
        t0 = time.perf_counter()
        interrupted = None
        try:
            mine = run_shard(args, rank, world, local_rank, record_dir)
        except Exception as exc:
            mine = {"error": repr(exc), "rank": rank}
        except (KeyboardInterrupt, SystemExit) as exc:     # reported like a failure, re-raised behind the collectives
            mine = {"error": repr(exc), "rank": rank}
            interrupted = exc
        gathered = [mine]
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
        failed = [s for s in gathered if "error" in s]
        good = [s for s in gathered if "error" not in s]
        result = None
        if good:
            elapsed = max(time.perf_counter() - t0, max(s["seconds"] for s in good))
            result = aggregate(good, elapsed, args.visits, args.boards)
            if rank == 0:
                report(result, args.json)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if interrupted is not None:
            raise interrupted
        if failed:
            raise SystemExit(f"self-play shard(s) failed: {failed}")
        return result

    # stand-alone: this process is the launcher (selfplay_main.py:56-66), one child per GPU
    import multiprocessing as mp
    world = args.process
    if world <= 0:
        import torch
        world = max(1, torch.cuda.device_count())
    n = args.resume_dir or next_record_dir(args.save_dir)
    record_dir = os.path.join(args.save_dir, str(n))
    os.makedirs(record_dir, exist_ok=True)
    print(f"Self play visits : {args.visits}")
    argv = list(sys.argv[1:] if argv is None else argv)
    ctx = mp.get_context("spawn")                 # HIP state must not be forked
    t0 = time.perf_counter()
    procs, pipes = [], []
    for rank in range(world):
        parent, child = ctx.Pipe(duplex=False)
        p = ctx.Process(target=_spawn_entry, args=(argv, rank, world, record_dir, child), name=f"selfplay-shard-{rank}")
        p.start()
        child.close()
        procs.append(p)
        pipes.append(parent)
    per_rank = []
    for p, pipe in zip(procs, pipes):
        try:
            per_rank.append(pipe.recv())
        except EOFError:
            per_rank.append({"error": "shard exited without a result", "rank": len(per_rank)})
        p.join()
    failed = [s for s in per_rank if "error" in s]
    if failed:
        raise SystemExit(f"self-play shard(s) failed: {failed}")
    result = aggregate(per_rank, time.perf_counter() - t0, args.visits, args.boards)
    report(result, args.json)
    return result


if __name__ == "__main__":
    main()
