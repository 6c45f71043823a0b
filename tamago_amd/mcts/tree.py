"""MCTSTree on MI355X - the reference's search API (mcts/tree.py:26-105,318-356,424-430)
over the device-resident search engine.

``MCTSTree(network, tree_size, batch_size, cgos_mode)`` / ``search_best_move`` / ``search``
/ ``get_root`` keep the reference's signatures and semantics: the caller's board is never
modified, the tree is rebuilt on every call, the return value is a padded-board
coordinate (PASS = 0, RESIGN = -1), and the Dirichlet / Gumbel draws come out of numpy's
GLOBAL legacy generator, whose state is advanced exactly as the reference would advance it.
"""
from typing import Any, Dict

import numpy as np

from tamago_amd.board.constant import PASS, RESIGN
from tamago_amd.board.go_board import GoBoard
from tamago_amd.board.stone import Stone, color_value
from tamago_amd.mcts.batch_data import BatchQueue
from tamago_amd.mcts.constant import MCTS_TREE_SIZE, NN_BATCH_SIZE, RESIGN_THRESHOLD, \
    PLAYOUTS, MAX_CONSIDERED_NODES
from tamago_amd.mcts.sequential_halving import get_candidates_and_visit_pairs
from tamago_amd.mcts.engine import SearchEngine, HostEvaluator, DeviceEvaluator
from tamago_amd.mcts.node import MCTSNode
from tamago_amd.mcts.time_manager import TimeControl, TimeManager


class _NodeList:
    """``tree.node[i]`` -> host snapshot of node i (negative indices as in a list)."""

    def __init__(self, tree):
        self._tree = tree

    def __len__(self):
        return self._tree.tree_size

    def __getitem__(self, index: int) -> MCTSNode:
        if index < 0:
            index += self._tree.tree_size
        return self._tree._engine_for(None).read_node(0, index)


class MCTSTree:
    def __init__(self, network, tree_size: int = MCTS_TREE_SIZE, batch_size: int = NN_BATCH_SIZE,
                 cgos_mode: bool = False, device_index: int = 0):
        self.network = network
        self.tree_size = tree_size
        self.batch_size = batch_size
        self.cgos_mode = cgos_mode
        self.device_index = device_index
        self.num_nodes = 0
        self.root = 0
        self.current_root = 0
        self.to_move = Stone.BLACK
        self.node = _NodeList(self)
        self.ponder_max_nodes = 1 << 22            # 4 M nodes = 13 GB of 9x9 pool: cap of ponder's growth
        self._engine = None
        self._engine_key = None
        self._gumbel_root = False

    @property
    def batch_queue(self) -> BatchQueue:
        """mcts/batch_data.py:7-34 view of the DEVICE leaf queue (input planes, paths, node indices
        of the leaves waiting for the network).  Like the reference's, it is empty whenever a
        search call has returned (every mini-batch is flushed before, tree.py:150-152,315)."""
        if self._engine is None:
            return BatchQueue()
        return self._engine.read_queue(0)

    # ------------------------------------------------------------------------------------
    def _evaluator(self):
        from tamago_amd.nn.network.dual_net import DualNet
        import torch
        if isinstance(self.network, DualNet):
            return DeviceEvaluator(self.network)
        return HostEvaluator(self.network, torch.device("cuda", self.device_index))

    def _engine_for(self, board, batch_size=None):
        if board is None:
            if self._engine is None:
                raise RuntimeError("no search has run yet")
            return self._engine
        batch = batch_size or self.batch_size
        net_size = getattr(self.network, "board_size", None)
        if net_size is not None and net_size != board.board_size:
            raise ValueError(f"network is built for {net_size}x{net_size}, board is "
                             f"{board.board_size}x{board.board_size}")
        key = (board.board_size, bool(board.check_superko), batch, self.cgos_mode, self.tree_size)
        if self._engine is None or key != self._engine_key:
            if self._engine is not None:
                self._engine.close()
            self._engine = SearchEngine(board.board_size, 1, self.tree_size, batch,
                                        self._evaluator(), self.cgos_mode, board.check_superko,
                                        self.device_index)
            self._engine_key = key
        return self._engine

    def _commit_rng(self, engine):
        """Leave numpy's global generator where the reference would have left it."""
        np.random.set_state(engine.streams[0].final_state())

    def get_root(self) -> MCTSNode:
        root = self._engine_for(None).read_node(0, self.current_root)
        noise = getattr(self._engine, "noise", None)
        if noise is not None and self._gumbel_root:
            root.noise = noise[0].copy()
        return root

    # ------------------------------------------------------------------------------------
    def search_best_move(self, board: GoBoard, color, time_manager: TimeManager,
                         analysis_query: Dict[str, Any] = None) -> int:
        """mcts/tree.py:57-105.  The reference doubles its node list when it fills up
        (tree.py:254-258); the device pool is grown in place the same way between mini-batches
        (SearchEngine.ensure_capacity -> tg_search_grow), so a search is never repeated and the
        clock keeps running over a growth."""
        return self._search_best_move(board, color, time_manager, analysis_query)

    def _sync_size(self, engine):
        """tree_size follows the pool (len(tree.node) after the reference's doubling)."""
        if engine.N != self.tree_size:
            self.tree_size = engine.N
            key = self._engine_key
            self._engine_key = key[:4] + (engine.N,)

    def _search_best_move(self, board, color, time_manager, analysis_query):
        engine = self._engine_for(board)
        self._gumbel_root = False
        self.to_move = color if isinstance(color, Stone) else Stone(color_value(color))
        engine.set_root(0, board, color, np.random.get_state())
        engine.root_eval(use_logit=False)                              # _initialize_search
        time_manager.start_timer()
        if int(engine.root_children[0]) == 1:                              # tree.py:76-77 (the root's child count off the draw cursor)
            self.num_nodes = int(engine.num_nodes()[0])
            self._commit_rng(engine)
            return PASS
        self.search(board, color, time_manager, analysis_query or {}, _engine=engine)
        root = engine.read_node(0, 0)
        self.num_nodes = root.tree_num_nodes
        self._sync_size(engine)
        self._commit_rng(engine)
        search_time = time_manager.calculate_consumption_time()
        time_manager.set_search_speed(root.node_visits, max(search_time, 1e-9))
        time_manager.substract_consumption_time(color, search_time)
        next_index = root.get_best_move_index()
        if root.calculate_value_evaluation(next_index) < RESIGN_THRESHOLD:
            return RESIGN
        return root.action[next_index]

    def search(self, board: GoBoard, color, time_manager: TimeManager,
               analysis_query: Dict[str, Any] = None, _engine=None):
        """mcts/tree.py:130-152: `threshold` descents in mini-batches of batch_size; the
        early-stop test (time_manager.py:135-163) runs after every mini-batch, which is
        where its inputs change."""
        engine = _engine or self._engine_for(board)
        threshold = time_manager.get_num_visits_threshold(color)
        done = 0
        # tree.py:168-174 prints the final analysis (interval 0) at the end of search() - BEFORE search_best_move flushes the
        # last, partial mini-batch (tree.py:84-85): its leaves are selected (virtual losses in place) but not yet evaluated
        # or backed up, and the printed visit counts do not include them
        late_analysis = bool(analysis_query) and analysis_query.get("interval", 0) == 0
        pending = False
        if threshold > 0 and time_manager.mode == TimeControl.STRICT_PLAYOUT and not analysis_query and engine.can_chain(threshold):
            # nothing between the mini-batches depends on their results (no early stop, the 10 000 s limit never ends it): queue them all
            batches = [self.batch_size] * (threshold // self.batch_size)
            if threshold % self.batch_size:
                batches.append(threshold % self.batch_size)
            engine.puct_chain(batches)
            done = threshold
        while done < threshold:
            leaves = min(self.batch_size, threshold - done)
            engine.ensure_capacity(leaves)
            if late_analysis and leaves < self.batch_size and done + leaves == threshold:
                engine.puct_select(leaves)
                pending = True
            else:
                engine.puct_batch(leaves)
            done += leaves
            if leaves == self.batch_size and done < threshold:
                if time_manager.is_time_over():
                    break
                # STRICT_PLAYOUT never decides early (time_manager.py:160-161): no root read-back
                if time_manager.mode != TimeControl.STRICT_PLAYOUT and \
                        time_manager.is_move_decided(engine.read_node(0, 0), threshold):
                    break
        if analysis_query and analysis_query.get("interval", 0) == 0:      # tree.py:170-174
            import sys
            sys.stdout.write(engine.read_node(0, 0).get_analysis(board, analysis_query.get("mode", "lz"),
                                                                 self.get_pv_lists))
            sys.stdout.flush()
        if pending:
            engine.puct_flush()

    def ponder(self, board: GoBoard, color, analysis_query: Dict[str, Any]):
        """mcts/tree.py:108-127: search without a visit limit until input arrives on stdin
        (``analysis_query["ponder"]``), printing analysis every ``interval`` seconds.  The
        reference polls stdin after every descent; here the poll sits between mini-batches (a
        batch in flight is finished), input that is already waiting stops after one descent.
        The node pool doubles when the next mini-batch might not fit (tree.py:254-258), up to
        `ponder_max_nodes` nodes (the reference is bounded by host memory only)."""
        import select
        import sys
        import time
        engine = self._engine_for(board)
        self._gumbel_root = False
        self.to_move = color if isinstance(color, Stone) else Stone(color_value(color))
        engine.set_root(0, board, color, np.random.get_state())
        engine.root_eval(use_logit=False)                              # _initialize_search
        query = analysis_query or {}
        interval = query.get("interval", 0)
        mode = query.get("mode", "lz")
        clock = time.time()

        def stdin_ready():
            if not query.get("ponder", False):
                return False
            ready, _, _ = select.select([sys.stdin], [], [], 0)
            return bool(ready)

        if engine.read_node(0, 0).get_num_children() > 1:
            while True:
                if engine.node_bound + self.batch_size > engine.N:
                    engine.node_bound = int(engine.num_nodes()[0])
                    if engine.node_bound + self.batch_size > engine.N and \
                            engine.N * 2 > self.ponder_max_nodes:
                        break
                engine.ensure_capacity(self.batch_size)
                waiting = stdin_ready()
                engine.puct_batch(1 if waiting else self.batch_size)
                if waiting or stdin_ready():
                    break
                if query and interval > 0 and time.time() - clock > interval:
                    clock = time.time()
                    sys.stdout.write(engine.read_node(0, 0).get_analysis(board, mode, self.get_pv_lists))
                    sys.stdout.flush()
        if query and interval == 0:                                    # tree.py:170-174
            sys.stdout.write(engine.read_node(0, 0).get_analysis(board, mode, self.get_pv_lists))
            sys.stdout.flush()
        self.num_nodes = int(engine.num_nodes()[0])
        self._sync_size(engine)
        self._commit_rng(engine)

    def search_with_callback(self, board: GoBoard, color, callback):
        """mcts/tree.py:177-196: one descent at a time (mini-batches of one leaf); the callback
        gets the descent's [(node index, child index), ...] and ends the search by returning True."""
        engine = self._engine_for(board, batch_size=1)
        self._gumbel_root = False
        self.to_move = color if isinstance(color, Stone) else Stone(color_value(color))
        engine.set_root(0, board, color, np.random.get_state())
        engine.root_eval(use_logit=False)
        while True:
            engine.ensure_capacity(1)
            engine.puct_batch(1)
            if callback(engine.read_path(0, 0)):
                break
        self.num_nodes = int(engine.num_nodes()[0])
        self._sync_size(engine)
        self._commit_rng(engine)

    # ---- principal variations (mcts/tree.py:432-473) -----------------------------------------
    def get_pv_lists(self, root: MCTSNode, coord) -> Dict[str, Any]:
        pv = {}
        for i in range(root.num_children):
            if root.children_visits[i] > 0:
                seq = self.get_best_move_sequence([root.action[i]], int(root.children_index[i]))
                pv[coord.convert_to_gtp_format(root.action[i])] = \
                    [coord.convert_to_gtp_format(p) for p in seq]
        return pv

    def get_best_move_sequence(self, pv_list, index: int):
        node = self.node[index]
        if node.node_visits == 0:
            return pv_list
        best = node.get_best_move_index()
        pv_list.append(node.action[best])
        nxt = int(node.children_index[best])
        if nxt == -1:
            return pv_list
        return self.get_best_move_sequence(pv_list, nxt)

    # ------------------------------------------------------------------------------------
    def generate_move_with_sequential_halving(self, board: GoBoard, color, time_manager: TimeManager,
                                              never_resign: bool) -> int:
        """mcts/tree.py:318-356 (Gumbel AlphaZero root + sequential halving)."""
        import time as _time
        start = _time.time()
        visits = time_manager.get_num_visits_threshold(color)
        # every phase is one mini-batch of num_considered * max_count leaves (<= visits)
        engine = self._engine_for(board, batch_size=max(visits, 1))
        self._gumbel_root = True
        engine.set_root(0, board, color, np.random.get_state())
        engine.root_eval(use_logit=True)
        engine.set_gumbel_noise()
        nc = engine.root_children                  # (child count off the draw cursor: no read-back behind the root evaluation)
        base = int(nc[0]) if nc[0] < MAX_CONSIDERED_NODES else MAX_CONSIDERED_NODES
        for num_considered, max_count in get_candidates_and_visit_pairs(base, visits).items():
            engine.ensure_capacity(num_considered * max_count)
            engine.gumbel_phase([num_considered], [max_count])
        root = self.get_root()
        self.num_nodes = root.tree_num_nodes
        self._commit_rng(engine)
        next_index = root.select_move_by_sequential_halving_for_root(PLAYOUTS)
        value = root.calculate_value_evaluation(next_index)
        time_manager.set_search_speed(root.node_visits, max(_time.time() - start, 1e-9))
        if not never_resign and value < 0.05:
            return RESIGN
        return root.get_child_move(next_index)
