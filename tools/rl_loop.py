"""One or more generations of the reference's pipeline (pipeline.sh: self-play -> train; the
GNU Go adjudication step is out of scope) on one GPU, every stage on this repo's path:

    python tools/rl_loop.py <program_dir> [generations] [games] [boards] [visits] [batch]

  self-play   tamago_amd.selfplay.worker.selfplay_shard   (HIP search + forward, SGF records)
  data        tamago_amd.nn.data_generator                (HIP featurise, rl_data_*.npz)
  train       tamago_amd.nn.learn                         (fp32 step, rl-model.bin / rl-state.ckpt)
"""
import glob
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import tamago_amd.nn.data_generator as dg  # noqa: E402
from tamago_amd.nn import learn  # noqa: E402
from tamago_amd.nn.network.dual_net import DualNet  # noqa: E402
from tamago_amd.selfplay.worker import selfplay_shard  # noqa: E402


def run_generation(program_dir, generation, games, boards, visits, batch, size=9, log=print):
    device = torch.device("cuda", 0)
    model = os.path.join(program_dir, "model", "rl-model.bin")
    net = DualNet(device, size)
    if os.path.exists(model):
        net.load_state_dict(torch.load(model, map_location="cpu"))
    else:                                   # generation 0 starts from the random initialisation
        os.makedirs(os.path.dirname(model), exist_ok=True)
        torch.save(net.state_dict(), model)
    kifu_dir = os.path.join(program_dir, "archive", str(generation))
    os.makedirs(kifu_dir, exist_ok=True)
    first = generation * games + 1
    t0 = time.time()
    stats = selfplay_shard(kifu_dir, net, list(range(first, first + games)), size, visits, boards=boards)
    t1 = time.time()
    for old in glob.glob(os.path.join(program_dir, "data", "rl_data_*.npz")):
        os.remove(old)
    os.makedirs(os.path.join(program_dir, "data"), exist_ok=True)
    dg.generate_reinforcement_learning_data(program_dir, [kifu_dir], size)
    t2 = time.time()
    loss = learn.train_with_gumbel_alphazero_on_gpu(program_dir, size, batch)
    t3 = time.time()
    log(f"generation {generation}: self-play {stats['games']} games / {stats['leaf_evals']} leaf-evals "
        f"in {t1 - t0:.1f} s, data {t2 - t1:.1f} s, train {t3 - t2:.1f} s, last-chunk loss sums {loss}")
    return stats, loss


if __name__ == "__main__":
    a = sys.argv[1:]
    program_dir = a[0]
    gens, games, boards, visits, batch = (int(x) for x in (a[1:6] + ["2", "256", "256", "16", "256"][len(a) - 1:]))
    dg.BATCH_SIZE = batch
    for g in range(gens):
        run_generation(program_dir, g, games, boards, visits, batch)
