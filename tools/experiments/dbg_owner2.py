import os, sys, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from oracle.stubnet import StubNet
from tamago_amd.mcts.tree import MCTSTree
from tamago_amd.mcts.time_manager import TimeManager, TimeControl
from test_gpu_search import load_npz, load_json, product_replay
size = 9
brd = load_npz(f"board_s{size}.npz")
recs = [r for r in load_json(f"trees_s{size}.json") if r["kind"] == "puct"]
def run(rec, owner):
    os.environ[os.environ.get("DBG_VAR", "TG_SELECT_OWNER")] = "1" if owner else "0"
    board = product_replay(size, brd["g0_move"], brd["g0_color"], rec["ply"], rec["superko"])
    net = StubNet(salt=rec["seed"])
    tree = MCTSTree(net, tree_size=2048, batch_size=rec["batch"], cgos_mode=rec["cgos"])
    mode = TimeControl.STRICT_PLAYOUT if rec["mode"] == "STRICT" else TimeControl.CONSTANT_PLAYOUT
    np.random.seed(rec["seed"])
    tree.search_best_move(board, rec["color"], TimeManager(mode, rec["visits"]), {})
    nodes = []
    for i in range(tree.num_nodes):
        nd = tree.node[i]; n = nd.num_children
        nodes.append(dict(act=np.array(nd.action[:n]), idx=nd.children_index[:n].copy(), vis=nd.children_visits[:n].copy(), vl=nd.children_virtual_loss[:n].copy(),
                          vs=nd.children_value_sum[:n].copy(), pol=nd.children_policy[:n].copy(), nv=int(nd.node_visits), nvl=int(nd.virtual_loss)))
    return nodes
for ri, rec in enumerate(recs):
    x = run(rec, False)
    try:
        y = run(rec, True)
    except Exception as ex:
        print('rec', ri, 'owner failed:', ex); continue
    print("rec", ri, {k: rec[k] for k in ("ply", "batch", "visits", "mode", "cgos", "superko")}, "nodes", len(x), len(y))
    bad = 0
    for n, (p, q) in enumerate(zip(x, y)):
        for key in ("act", "idx", "vis", "vl", "vs", "pol"):
            if not np.array_equal(p[key], q[key]):
                d = np.flatnonzero(p[key] != q[key]) if len(p[key]) == len(q[key]) else []
                print(f"  node {n} {key}: edges {d[:6]} ref {p[key][d[:6]] if len(d) else len(p[key])} owner {q[key][d[:6]] if len(d) else len(q[key])}"); bad += 1
        if p["nv"] != q["nv"] or p["nvl"] != q["nvl"]:
            print(f"  node {n} node_visits/vl ref {p['nv']},{p['nvl']} owner {q['nv']},{q['nvl']}"); bad += 1
        if bad > 16: break
    print("  mismatches", bad)
