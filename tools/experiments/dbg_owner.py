import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle.stubnet import StubNet
from tamago_amd.board.go_board import GoBoard
from tamago_amd.mcts.engine import SearchEngine, HostEvaluator
size = int(sys.argv[1]) if len(sys.argv) > 1 else 9
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 6
def run(owner):
    os.environ[os.environ.get("DBG_VAR", "TG_SELECT_OWNER")] = "1" if owner else "0"
    eng = SearchEngine(size, 1, batch * nb + 16, batch, HostEvaluator(StubNet(3), torch.device("cuda:0")), check_superko=True)
    eng.set_root(0, GoBoard(size, 7.0, True), 1, np.random.RandomState(100).get_state())
    eng.root_eval(False)
    out = []
    for i in range(nb):
        eng.puct_batch(batch)
        nn = int(eng.num_nodes()[0])
        nodes = []
        for node in range(nn):
            nd = eng.read_node(0, node)
            n = nd.num_children
            nodes.append(dict(idx=nd.children_index[:n].copy(), vis=nd.children_visits[:n].copy(), vl=nd.children_virtual_loss[:n].copy(),
                              vs=nd.children_value_sum[:n].copy(), pol=nd.children_policy[:n].copy(), nv=int(nd.node_visits), nvl=int(nd.virtual_loss)))
        out.append(nodes)
    return out
a = run(False); b = run(True)
for i, (x, y) in enumerate(zip(a, b)):
    if len(x) != len(y): print("batch", i, "num_nodes", len(x), len(y)); break
    bad = 0
    for n, (p, q) in enumerate(zip(x, y)):
        for key in ("idx", "vis", "vl", "vs", "pol"):
            if not np.array_equal(p[key], q[key]):
                d = np.flatnonzero(p[key] != q[key])
                print(f"batch {i} node {n} {key}: edges {d[:6]} ref {p[key][d[:6]]} owner {q[key][d[:6]]}"); bad += 1
        if p["nv"] != q["nv"] or p["nvl"] != q["nvl"]:
            print(f"batch {i} node {n} node_visits/vl ref {p['nv']},{p['nvl']} owner {q['nv']},{q['nvl']}"); bad += 1
        if bad > 12: break
    print("batch", i, "nodes", len(x), "mismatches", bad)
    if bad: break
