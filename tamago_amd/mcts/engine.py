"""Lock-step driver of T device-resident search trees (host side of the C ABI in
include/tamago_hip.h).  MCTSTree (one tree, the reference's API) and the self-play worker
(many boards per GPU) are thin layers over this class.

Random streams: the reference consumes numpy's legacy MT19937 stream in program order -
n doubles per node expansion (Dirichlet(1..1) prior, mcts/tree.py:509-519) and A doubles
per Gumbel move (mcts/node.py:275-278).  ``ExpStream`` reproduces that stream position by
position with ``RandomState.standard_exponential`` / ``gumbel`` (the very C functions
numpy's dirichlet / gumbel use, so the doubles are bit-identical) and hands windows of it
to the device, which consumes them with a cursor.
"""
import ctypes
import math
import os
from typing import List, Optional

import numpy as np
import torch

from tamago_amd import lib as _lib
from tamago_amd.board.go_board import GoBoard, zobrist_keys
from tamago_amd.board.stone import color_value
from tamago_amd.mcts.node import MCTSNode


class ExpStream:
    """Position-addressable view of a legacy numpy stream as exponentials e_i = -log(1-u_i).

    Gumbel noise comes from the same values: numpy's legacy gumbel is
    ``loc - scale*log(-log(1-u))`` = ``-log(e_i)`` with the same libm ``log`` (verified
    bit-for-bit in tests/test_host_rng.py), so no second generator has to be positioned."""

    def __init__(self, state):
        self.origin = state                       # RandomState state at position 0
        self.gen = np.random.RandomState()        # generator at position buf0 + len(buf)
        self.gen.set_state(state)
        self.buf = np.empty(0, dtype=np.float64)  # exponentials for positions [buf0, buf0+len)
        self.buf0 = 0
        self.pos = 0                              # next unconsumed position

    def window(self, need: int) -> np.ndarray:
        have = self.buf0 + len(self.buf) - self.pos
        if have < need:
            extra = self.gen.standard_exponential(need - have)
            self.buf = np.concatenate([self.buf[self.pos - self.buf0:], extra])
            self.buf0 = self.pos
        off = self.pos - self.buf0
        return self.buf[off:off + need]

    def consume(self, count: int):
        self.pos += int(count)

    def gumbel(self, size: int) -> np.ndarray:
        """Next `size` positions as Gumbel(0,1) noise (node.py:278)."""
        noise = np.array([-math.log(v) for v in self.window(size)], dtype=np.float64)
        self.pos += size
        return noise

    def final_state(self):
        """Generator state after everything consumed so far (O(pos): per-move streams only)."""
        g = np.random.RandomState()
        g.set_state(self.origin)
        if self.pos:
            g.random_sample(self.pos)             # one double per position
        return g.get_state()


class LibStream:
    """Handle on a library-owned legacy stream (tg_search_seed_stream): the C side generates,
    stages and uploads the draws; Python only seeds it and reads the state back."""

    def __init__(self, engine, tree: int, state):
        self.engine = engine
        self.tree = tree
        key = np.ascontiguousarray(state[1], dtype=np.uint32)
        assert state[0] == "MT19937" and key.shape == (624,)
        _lib.check(engine.lib.tg_search_seed_stream(engine.handle, tree, key.ctypes.data, int(state[2])),
                   "tg_search_seed_stream")

    def final_state(self):
        """numpy state after everything this tree has consumed so far."""
        key = np.zeros(624, dtype=np.uint32)
        pos = ctypes.c_int(0)
        _lib.check(self.engine.lib.tg_search_stream_state(self.engine.handle, self.tree, key.ctypes.data,
                                                          ctypes.byref(pos)), "tg_search_stream_state")
        return ("MT19937", key, int(pos.value), 0, 0.0)


class HostEvaluator:
    """Adapter for any object with the DualNet host API (inference /
    inference_with_policy_logits on CPU tensors): planes are copied to the host, results
    back to the device.  Used for stand-in evaluators in the parity tests and for drop-in
    compatibility with arbitrary networks."""

    def __init__(self, network, device):
        self.network = network
        self.device = device
        self.batches: List[int] = []

    def __call__(self, planes: torch.Tensor, want_logits: bool):
        x = planes.cpu()
        self.batches.append(x.shape[0])
        if want_logits:
            policy, value = self.network.inference_with_policy_logits(x)
        else:
            policy, value = self.network.inference(x)
        return (policy.to(self.device, torch.float32).contiguous(),
                value.to(self.device, torch.float32).contiguous())


class DeviceEvaluator:
    """DualNet forward without a host hop (tg_net_forward_dev)."""

    def __init__(self, network):
        self.network = network
        self.batches: List[int] = []

    def __call__(self, planes: torch.Tensor, want_logits: bool):
        self.batches.append(planes.shape[0])
        return self.network.forward_device(planes, want_logits)


class SearchEngine:
    def __init__(self, board_size: int, num_trees: int, tree_size: int, batch_size: int,
                 evaluator, cgos_mode: bool = False, check_superko: bool = False,
                 device_index: int = 0, host_streams: bool = False):
        """host_streams: generate the random draws in Python (ExpStream + tg_search_set_rng /
        tg_search_rng_consumed / tg_search_set_noise - the calls a numpy-side host would make)
        instead of the library-owned streams; same results, kept for that boundary's tests."""
        self.lib = _lib.load()
        self.S = board_size
        self.P = board_size * board_size
        self.A = self.P + 1
        self.T = num_trees
        self.N = tree_size
        self.K = batch_size
        self.device = torch.device("cuda", device_index)
        self.evaluator = evaluator
        cfg = _lib.SearchConfig(board_size, num_trees, tree_size, batch_size, int(cgos_mode),
                                int(check_superko), device_index, 0)
        handle = ctypes.c_void_p()
        _lib.check(self.lib.tg_search_create(ctypes.byref(cfg), ctypes.byref(handle)),
                   "tg_search_create")
        self.handle = handle
        keys = zobrist_keys(board_size)
        _lib.check(self.lib.tg_search_set_zobrist(self.handle, keys.ctypes.data, keys.size),
                   "tg_search_set_zobrist")
        self.planes = torch.empty((num_trees * batch_size, 6, board_size, board_size),
                                  dtype=torch.float32, device=self.device)
        self.host_streams = host_streams or bool(os.environ.get("TG_HOST_STREAMS"))
        self.streams: List[Optional[ExpStream]] = [None] * num_trees
        self._pool = None
        self.node_bound = 0                       # upper bound of nodes in use in any tree
        self._window_left = 0
        self._window_cap = 0
        self._window_used = np.zeros(num_trees, dtype=np.int64)

    def close(self):
        if self.handle is not None:
            self.lib.tg_search_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # ---------------------------------------------------------------------------------
    def set_root(self, tree: int, board: GoBoard, color, rng_state=None):
        """Position + (optionally) the legacy-RNG state this tree draws from."""
        assert board.board_size == self.S
        cells = np.ascontiguousarray(board.cells, dtype=np.uint8)
        hist = np.ascontiguousarray(board.rec_hash[:min(board.moves, board.max_records)])
        pos = _lib.RootPosition(cells.ctypes.data, hist.ctypes.data,
                                ctypes.c_uint64(int(board.hash)), board.moves, board.ko_pos,
                                board.ko_move, board.prev_move(1), board.prev_move(2),
                                color_value(color))
        _lib.check(self.lib.tg_search_set_root(self.handle, tree, ctypes.byref(pos)),
                   "tg_search_set_root")
        if rng_state is not None:
            self.set_stream(tree, rng_state)

    def set_stream(self, tree: int, rng_state):
        """(Re)seed the legacy stream tree `tree` draws from (np.random.get_state() layout)."""
        self.streams[tree] = ExpStream(rng_state) if self.host_streams else LibStream(self, tree, rng_state)
        self._window_left = 0

    def _feed_rng(self, need: int):
        """Upload the next `need` stream positions of every tree (window becomes active at the
        next root / select launch).  Skipped when an earlier (pre-fetched) window still covers
        `need` positions for every tree."""
        if not self.host_streams:
            _lib.check(self.lib.tg_search_feed_streams(self.handle, need, int(self._window_left == 0)),
                       "tg_search_feed_streams")
            self._window_left = -1                   # the library tracks the window from here on
            return
        if self._window_left >= need:
            return
        self._window_left = need
        self._window_cap = need
        self._window_used = np.zeros(self.T, dtype=np.int64)
        win = np.empty((self.T, need), dtype=np.float64)

        def fill(t):
            win[t] = self.streams[t].window(need)

        if self.T >= 8:
            # numpy's legacy generators release the GIL while filling, so threads scale
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1))
            list(self._pool.map(fill, range(self.T)))
        else:
            for t in range(self.T):
                fill(t)
        _lib.check(self.lib.tg_search_set_rng(self.handle, win.ctypes.data, need, need),
                   "tg_search_set_rng")

    def _collect_rng(self):
        if not self.host_streams:
            delta = np.zeros(self.T, dtype=np.int64)
            _lib.check(self.lib.tg_search_advance_streams(self.handle, delta.ctypes.data),
                       "tg_search_advance_streams")
            return delta
        used = np.zeros(self.T, dtype=np.int64)
        _lib.check(self.lib.tg_search_rng_consumed(self.handle, used.ctypes.data),
                   "tg_search_rng_consumed")
        # the device cursor is cumulative within the active window
        delta = used - self._window_used
        self._window_used = used
        for s, c in zip(self.streams, delta):
            s.consume(int(c))
        self._window_left = self._window_cap - int(used.max()) if len(used) else 0
        return delta

    def _evaluate_and_backup(self, n_slots: int, use_logit: bool, packed_total: int = 0):
        """Forward pass over the queued leaves + write-back / backup.  `packed_total` > 0: the
        leaves of the trees lie back to back (n_slots is ignored, 0 is passed to the library)."""
        planes = self.planes[:packed_total] if packed_total else self.planes[:self.T * n_slots]
        policy, value = self.evaluator(planes, use_logit)
        _lib.check(self.lib.tg_search_backup(self.handle, policy.data_ptr(), value.data_ptr(),
                                             0 if packed_total else n_slots, int(use_logit), self._stream()),
                   "tg_search_backup")
        self._keep = (policy, value)        # keep alive until the stream has consumed them

    def root_eval(self, use_logit: bool = False, first_batch: int = 0):
        """tree.py:49-54: expand + evaluate the root of every tree (one leaf each).  With
        `first_batch` the random window also covers the first mini-batch of that many leaves,
        so no upload sits between the (tiny) root evaluation and the first selection."""
        self.node_bound = 1
        self._queue_stride = 1
        self._feed_rng(self.A * (1 + first_batch))
        _lib.check(self.lib.tg_search_root_planes(self.handle, self.planes.data_ptr(),
                                                  self._stream()), "tg_search_root_planes")
        # a Dirichlet prior takes one draw per child (mcts/tree.py:509-519): what the root expansion consumed IS the roots'
        # child counts - known as soon as the expansion kernel is done, without waiting for the root's forward pass and backup
        self.root_children = np.asarray(self._collect_rng(), dtype=np.int64).copy()
        self._evaluate_and_backup(1, use_logit)

    def prefetch_rng(self, leaves: int):
        """Generate + upload the window for the NEXT root evaluation and its first mini-batch
        now (e.g. while the last forward pass of the current search is still running)."""
        self._window_left = 0
        self._feed_rng(self.A * (1 + leaves))

    def puct_batch(self, leaves: int):
        """`leaves` PUCT descents per tree + one evaluation + backup (tree.py:146-152 with
        the flush of :240-241)."""
        # order matters for overlap: the random window of THIS batch is generated and
        # uploaded (private copy stream) while the forward pass of the PREVIOUS batch is
        # still running; the cursor read-back waits for the selection kernel only
        self.node_bound += leaves
        self._queue_stride = leaves
        self._feed_rng(leaves * self.A)
        _lib.check(self.lib.tg_search_select_puct(self.handle, leaves, self.planes.data_ptr(),
                                                  None, self._stream()), "tg_search_select_puct")
        # forward + backup are queued BEFORE the cursor read-back: that read waits for the selection kernel only
        # (its own event on the copy stream), so the host is back while the forward pass runs
        self._evaluate_and_backup(leaves, False)
        self._collect_rng()

    def puct_chain(self, batches):
        """Several mini-batches queued in one go: ONE random window for all of them (the device cursor runs on from launch to
        launch), every selection / forward / backup launch enqueued back to back, one cursor read-back at the end - for searches
        whose course does not depend on what a mini-batch found (STRICT_PLAYOUT: no early stop, time_manager.py:160-161).  Same
        launches on the same data as puct_batch called once per mini-batch; what goes is the host's round trip between them
        (the selection launch of mini-batch k + 1 used to be queued only after the cursor of mini-batch k had been read back)."""
        arr = np.ascontiguousarray(batches, dtype=np.int32)
        total, kmax = int(arr.sum()), int(arr.max())
        self.node_bound += total
        self._queue_stride = int(arr[-1])
        policy = torch.empty((self.T * kmax, self.A), dtype=torch.float32, device=self.device)
        value = torch.empty((self.T * kmax, 3), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.tg_search_puct_chain(self.handle, self.evaluator.network.handle, arr.ctypes.data, len(arr),
                                                 int(self._window_left == 0), self.planes.data_ptr(), policy.data_ptr(),
                                                 value.data_ptr(), self._stream()), "tg_search_puct_chain")
        self._window_left = -1                       # the library tracks the window from here on
        self.evaluator.batches.extend(int(self.T * k) for k in arr)
        self._keep = (policy, value)                 # keep alive until the stream has consumed them
        self._collect_rng()

    def can_chain(self, total_leaves: int) -> bool:
        """puct_chain needs the library's streams (one window for the whole search), the device evaluator (the forward pass is
        queued by the library) and a pool that holds the whole search without growing (the reference doubles its node list
        between mini-batches, mcts/tree.py:254-258: that stays there)."""
        # (exactly DeviceEvaluator: the chained forward passes are launched by the library on the network handle and never go
        # through evaluator.__call__ - a subclass that wraps the forward pass keeps the per-mini-batch loop)
        return total_leaves > 0 and (not self.host_streams) and type(self.evaluator) is DeviceEvaluator and \
            hasattr(self.evaluator.network, "handle") and self.node_bound + total_leaves <= self.N

    def puct_select(self, leaves: int):
        """The selection half of puct_batch: afterwards the device queue holds `leaves` leaves per
        tree (read_queue) until puct_flush() evaluates and backs them up."""
        self.node_bound += leaves
        self._queue_stride = leaves
        self._feed_rng(leaves * self.A)
        _lib.check(self.lib.tg_search_select_puct(self.handle, leaves, self.planes.data_ptr(),
                                                  None, self._stream()), "tg_search_select_puct")
        self._collect_rng()

    def puct_flush(self):
        """process_mini_batch (tree.py:273-315) for the leaves queued by puct_select."""
        self._evaluate_and_backup(self._queue_stride, False)

    def ensure_capacity(self, leaves: int) -> bool:
        """Make room for `leaves` more descents (each allocates at most one node) in every tree:
        the reference doubles its node list when it fills up (mcts/tree.py:254-258), the device
        pool is grown in place the same way (tg_search_grow keeps the trees).  The host-side
        bound makes this free while the pool is far from full.  Returns True if the pool grew."""
        if self.node_bound + leaves <= self.N:
            return False
        self.node_bound = int(self.num_nodes().max())
        if self.node_bound + leaves <= self.N:
            return False
        new_size = self.N
        while self.node_bound + leaves > new_size:
            import sys
            sys.stderr.write(f"Tree is full. Allocate new space {new_size} -> {new_size * 2}\n")
            new_size *= 2
        _lib.check(self.lib.tg_search_grow(self.handle, new_size), "tg_search_grow")
        self.N = new_size
        return True

    def read_queue(self, tree: int = 0):
        """The device leaf queue of `tree` as a BatchQueue (mcts/batch_data.py:7-34): input planes,
        root-first paths and node indices of the leaves the last selection launch queued.  Empty
        once the mini-batch has been backed up (the reference clears its queue there, tree.py:315)."""
        from tamago_amd.mcts.batch_data import BatchQueue
        idx = np.zeros(self.K, dtype=np.int32)
        n = ctypes.c_int32(0)
        _lib.check(self.lib.tg_search_read_queue(self.handle, tree, idx.ctypes.data, self.K, ctypes.byref(n)),
                   "tg_search_read_queue")
        queue = BatchQueue()
        if n.value:
            planes = self.planes[tree * self._queue_stride:tree * self._queue_stride + n.value].cpu().numpy()
            for k in range(n.value):
                queue.push(planes[k], self.read_path(tree, k), int(idx[k]))
        return queue

    def play(self, moves):
        """Play one move per tree on the device-resident root positions (-1 = skip, -2 = the most visited root child's
        move, chosen on the device)."""
        mv = np.ascontiguousarray(moves, dtype=np.int32)
        assert mv.shape == (self.T,)
        _lib.check(self.lib.tg_search_play(self.handle, mv.ctypes.data, self._stream()), "tg_search_play")

    def read_positions(self):
        """(cells [T][(S+2)^2], moves [T], to_move [T]) of the current root positions."""
        cells = np.zeros((self.T, (self.S + 2) ** 2), dtype=np.uint8)
        moves = np.zeros(self.T, dtype=np.int32)
        to_move = np.zeros(self.T, dtype=np.int32)
        _lib.check(self.lib.tg_search_read_positions(self.handle, cells.ctypes.data, moves.ctypes.data,
                                                     to_move.ctypes.data), "tg_search_read_positions")
        return cells, moves, to_move

    def set_gumbel_noise(self):
        """node.py:275-278 for every root: A doubles from each tree's stream, drawn after the
        root's Dirichlet prior and NN evaluation (tree.py:332-336)."""
        noise = np.empty((self.T, self.A), dtype=np.float64)
        if not self.host_streams:
            _lib.check(self.lib.tg_search_draw_noise(self.handle, noise.ctypes.data), "tg_search_draw_noise")
            self.noise = noise
            return noise
        self._window_left = 0                     # the noise sits between two windows
        for t, s in enumerate(self.streams):
            noise[t] = s.gumbel(self.A)
        _lib.check(self.lib.tg_search_set_noise(self.handle, noise.ctypes.data), "tg_search_set_noise")
        self.noise = noise
        return noise

    def gumbel_phase(self, num_considered, max_count, packed: bool = True):
        """One sequential-halving phase for every tree (tree.py:375-384): per-tree
        (num_considered, max_count), one evaluation of all queued leaves, backup."""
        nc = np.ascontiguousarray(num_considered, dtype=np.int32)
        mc = np.ascontiguousarray(max_count, dtype=np.int32)
        per_tree = nc.astype(np.int64) * mc
        slots = int(per_tree.max())
        if slots == 0:
            return
        # packed leaf layout (slots_per_tree = 0): the forward pass covers exactly the queued
        # leaves - a tree whose root has one candidate runs 1 x `visits` levels and would
        # otherwise stretch every tree's slot range to `visits`
        self.node_bound += slots
        self._feed_rng(slots * self.A)
        _lib.check(self.lib.tg_search_select_gumbel(self.handle, nc.ctypes.data, mc.ctypes.data,
                                                    0 if packed else slots,
                                                    self.planes.data_ptr(), self._stream()),
                   "tg_search_select_gumbel")
        # forward + backup are queued before the cursor read-back (which waits for the selection kernel only)
        self._evaluate_and_backup(slots, True, packed_total=int(per_tree.sum()) if packed else 0)
        self._collect_rng()

    # ---------------------------------------------------------------------------------
    def num_nodes(self) -> np.ndarray:
        out = np.zeros(self.T, dtype=np.int32)
        _lib.check(self.lib.tg_search_num_nodes(self.handle, out.ctypes.data), "tg_search_num_nodes")
        return out

    def read_path(self, tree: int, slot: int):
        """[(node index, child index), ...] root first for queued leaf `slot` (tree.py:199-244 path)."""
        cap = 4 * self.P + 8
        nodes = np.zeros(cap, dtype=np.int32)
        edges = np.zeros(cap, dtype=np.int32)
        n = ctypes.c_int32(0)
        _lib.check(self.lib.tg_search_read_path(self.handle, tree, slot, nodes.ctypes.data, edges.ctypes.data,
                                                cap, ctypes.byref(n)), "tg_search_read_path")
        return [(int(nodes[i]), int(edges[i])) for i in range(n.value)]

    def read_roots(self):
        """(num_children [T], action [T][A], children_visits [T][A]) of every root."""
        nc = np.zeros(self.T, dtype=np.int32)
        action = np.zeros((self.T, self.A), dtype=np.int32)
        visits = np.zeros((self.T, self.A), dtype=np.int32)
        _lib.check(self.lib.tg_search_read_roots(self.handle, nc.ctypes.data, action.ctypes.data,
                                                 visits.ctypes.data), "tg_search_read_roots")
        return nc, action, visits

    def read_root_stats(self):
        """dict of root arrays for all trees ([T] / [T][A]) in one call."""
        t, a = self.T, self.A
        out = {"num_children": np.zeros(t, np.int32), "node_visits": np.zeros(t, np.int32),
               "raw_value": np.zeros(t, np.float32), "action": np.zeros((t, a), np.int32),
               "children_visits": np.zeros((t, a), np.int32),
               "children_virtual_loss": np.zeros((t, a), np.int32),
               "children_value_sum": np.zeros((t, a), np.float64),
               "children_policy": np.zeros((t, a), np.float64)}
        _lib.check(self.lib.tg_search_read_root_stats(
            self.handle, out["num_children"].ctypes.data, out["node_visits"].ctypes.data,
            out["raw_value"].ctypes.data, out["action"].ctypes.data,
            out["children_visits"].ctypes.data, out["children_virtual_loss"].ctypes.data,
            out["children_value_sum"].ctypes.data, out["children_policy"].ctypes.data),
            "tg_search_read_root_stats")
        return out

    def root_view(self, stats, tree: int) -> MCTSNode:
        """MCTSNode view of one root out of read_root_stats() (no further device access)."""
        view = MCTSNode(self.A)
        view.num_children = int(stats["num_children"][tree])
        view.node_visits = int(stats["node_visits"][tree])
        view.raw_value = np.float32(stats["raw_value"][tree])
        view.action = [int(x) for x in stats["action"][tree]]
        view.children_visits = stats["children_visits"][tree]
        view.children_virtual_loss = stats["children_virtual_loss"][tree]
        view.children_value_sum = stats["children_value_sum"][tree]
        view.children_policy = stats["children_policy"][tree]
        noise = getattr(self, "noise", None)
        if noise is not None:
            view.noise = noise[tree]
        return view

    def read_node(self, tree: int, node: int) -> MCTSNode:
        a = self.A
        view = MCTSNode(a)
        nc = ctypes.c_int32()
        nv = ctypes.c_int32()
        nvl = ctypes.c_int32()
        nvs = ctypes.c_float()
        raw = ctypes.c_float()
        action = np.zeros(a, dtype=np.int32)
        _lib.check(self.lib.tg_search_read_node(
            self.handle, tree, node, ctypes.byref(nc), ctypes.byref(nv), ctypes.byref(nvl),
            action.ctypes.data,
            view.children_index.ctypes.data, view.children_visits.ctypes.data,
            view.children_virtual_loss.ctypes.data, view.children_value_sum.ctypes.data,
            view.children_policy.ctypes.data, view.children_value.ctypes.data,
            ctypes.byref(nvs), ctypes.byref(raw)), "tg_search_read_node")
        view.num_children = nc.value
        view.node_visits = nv.value
        view.virtual_loss = nvl.value
        view.node_value_sum = np.float32(nvs.value)
        view.raw_value = np.float32(raw.value)
        view.action = [int(v) for v in action]
        n = ctypes.c_int32(0)
        _lib.check(self.lib.tg_search_node_record_num_nodes(self.handle, ctypes.byref(n)), "tg_search_node_record_num_nodes")
        view.tree_num_nodes = int(n.value)           # num_nodes of the tree, read with the node
        return view
