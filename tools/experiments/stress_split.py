"""Stress of the two-workgroup selection kernel: the same searches with TG_SELECT_SPLIT=0 / 1, many roots and batch
shapes, whole-tree digests compared.  python tools/experiments/stress_split.py [rounds]"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle.stubnet import StubNet
from tamago_amd.board.go_board import GoBoard
from tamago_amd.mcts.engine import SearchEngine, HostEvaluator
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
def digest(size, T, batch, nb, seed, split):
    os.environ["TG_SELECT_SPLIT"] = "1" if split else "0"
    eng = SearchEngine(size, T, batch * nb + 16, batch, HostEvaluator(StubNet(seed), torch.device("cuda:0")), check_superko=True)
    rs = np.random.RandomState(seed)
    for t in range(T):
        b = GoBoard(size, 7.0, True); c = 1
        for _ in range(rs.randint(0, 40 if size == 9 else 80)):
            for _try in range(50):
                pos = b.onboard_pos[rs.randint(len(b.onboard_pos))]
                if b.is_legal(pos, c): break
            else:
                break
            b.put_stone(pos, c); c = 3 - c
        eng.set_root(t, b, c, np.random.RandomState(1000 * seed + t).get_state())
    eng.root_eval(False)
    for i in range(nb):
        eng.puct_batch(batch if i < nb - 1 else max(1, batch - 5))
    h = hashlib.sha256()
    nn = eng.num_nodes(); h.update(nn.tobytes())
    for t in range(T):
        for node in range(0, int(nn[t]), 3):
            nd = eng.read_node(t, node)
            n = nd.num_children
            for arr in (nd.children_index[:n], nd.children_visits[:n], nd.children_virtual_loss[:n], nd.children_value_sum[:n], nd.children_policy[:n]):
                h.update(np.ascontiguousarray(arr).tobytes())
    eng.close() if hasattr(eng, "close") else None
    return h.hexdigest()[:16]
bad = 0
cfgs = [(9, 1, 256, 4), (9, 3, 64, 8), (9, 16, 32, 5), (19, 1, 64, 12), (9, 1, 8, 30), (9, 2, 256, 6), (19, 2, 32, 10)]
for r in range(rounds):
    for (size, T, batch, nb) in cfgs:
        a = digest(size, T, batch, nb, 7 * r + 1, False)
        b = digest(size, T, batch, nb, 7 * r + 1, True)
        ok = a == b
        bad += not ok
        print(f"round {r} size {size} trees {T} batch {batch} x {nb}: {'ok' if ok else 'MISMATCH'} {a} {b}", flush=True)
print("mismatches", bad)
