"""Oracle batched MCTS (TEST INFRASTRUCTURE - see oracle/__init__.py).

Restates mcts/tree.py:29-422 (PUCT search, batched leaf evaluation, Gumbel /
sequential-halving search), mcts/batch_data.py:7-34 and mcts/time_manager.py:61-163.
The Dirichlet "tentative" prior (tree.py:509-519) and the Gumbel noise (node.py:278)
are drawn from numpy's GLOBAL legacy RNG in program order, exactly as the reference
does, so a fixed ``np.random.seed`` reproduces the reference's visit counts.
"""
import time
from enum import Enum
from typing import List, Tuple

import numpy as np
import torch

from oracle.board import GoBoard, PASS, RESIGN, opponent, BLACK, WHITE
from oracle.feature import generate_input_planes
from oracle.halving import candidates_and_visit_pairs
from oracle.node import Node, NOT_EXPANDED

PLAYOUTS = 100              # mcts/constant.py:11
NN_BATCH_SIZE = 1           # mcts/constant.py:14
MAX_CONSIDERED_NODES = 16   # mcts/constant.py:23
RESIGN_THRESHOLD = 0.05     # mcts/constant.py:38
MCTS_TREE_SIZE = 65536      # mcts/constant.py:41
CONST_VISITS = 1000
CONST_TIME = 5.0
REMAINING_TIME = 60.0
VISITS_PER_SEC = 20


class TimeControl(Enum):
    """time_manager.py:12-18."""
    CONSTANT_PLAYOUT = 0
    CONSTANT_TIME = 1
    TIME_CONTROL = 2
    STRICT_PLAYOUT = 3


class TimeManager:
    """time_manager.py:21-163 (the parts that decide how many descents run)."""

    def __init__(self, mode: TimeControl, constant_visits: int = CONST_VISITS,
                 constant_time: float = CONST_TIME, remaining_time: float = REMAINING_TIME):
        self.mode = mode
        self.constant_visits = constant_visits
        self.constant_time = constant_time
        self.search_speed = VISITS_PER_SEC
        self.remaining_time = [remaining_time] * 2
        self.time_limit = 0
        self.start_time = 0

    def set_search_speed(self, visits: int, consumption_time: float):
        self.search_speed = visits / consumption_time if visits > 0 else VISITS_PER_SEC

    def get_num_visits_threshold(self, color: int) -> int:
        """time_manager.py:61-83."""
        if self.mode in (TimeControl.CONSTANT_PLAYOUT, TimeControl.STRICT_PLAYOUT):
            self.time_limit = 10000.0
            return int(self.constant_visits)
        if self.mode == TimeControl.CONSTANT_TIME:
            self.time_limit = self.constant_time
            threshold = int(self.search_speed * self.constant_time)
            return threshold if threshold > 0 else 1
        remaining = self.remaining_time[0] if color == BLACK else self.remaining_time[1]
        self.time_limit = remaining / 10.0
        threshold = int(self.search_speed * self.time_limit)
        return threshold if threshold > 0 else 1

    def start_timer(self):
        self.start_time = time.time()

    def is_time_over(self) -> bool:
        """time_manager.py:135-143."""
        return time.time() - self.start_time > self.time_limit

    def is_move_decided(self, root: Node, threshold: int) -> bool:
        """time_manager.py:146-163: early stop once the runner-up cannot catch up;
        disabled in STRICT_PLAYOUT.  Sorted over all A slots, like the reference."""
        ordered = sorted(root.children_visits)
        remaining = threshold - root.node_visits
        cutoff = ordered[-1] - ordered[-2]
        if self.mode == TimeControl.STRICT_PLAYOUT:
            cutoff = 0
        return bool(remaining < cutoff)


class BatchQueue:
    """batch_data.py:7-34."""

    def __init__(self):
        self.clear()

    def push(self, input_plane: np.ndarray, path: List[Tuple[int, int]], node_index: int):
        self.input_plane.append(input_plane)
        self.path.append(path)
        self.node_index.append(node_index)

    def clear(self):
        self.input_plane = []
        self.path = []
        self.node_index = []


def tentative_policy(n: int) -> np.ndarray:
    """tree.py:509-519: Dirichlet(1,...,1) from the global legacy stream."""
    return np.random.dirichlet(alpha=np.ones(n))


class MCTSTree:
    """tree.py:26-46."""

    def __init__(self, network, board_size: int, tree_size: int = MCTS_TREE_SIZE,
                 batch_size: int = NN_BATCH_SIZE, cgos_mode: bool = False):
        self.num_actions = board_size * board_size + 1
        self.node = [Node(self.num_actions) for _ in range(tree_size)]
        self.num_nodes = 0
        self.network = network
        self.batch_queue = BatchQueue()
        self.current_root = 0
        self.batch_size = batch_size
        self.cgos_mode = cgos_mode
        self.batch_log = []          # oracle-only: sizes of the evaluated mini-batches
        self.eval_hook = None        # oracle-only: called with (planes, policy, value, use_logit)

    # ------------------------------------------------------------------------------------
    def expand_node(self, board: GoBoard, color: int) -> int:
        """tree.py:247-270: pool doubles when full; candidates = legal, self-atari < 7,
        not a complete eye, PASS last; prior = Dirichlet draw."""
        index = self.num_nodes
        if index >= len(self.node):
            self.node.extend([Node(self.num_actions) for _ in range(len(self.node))])
        candidates = board.search_candidates(color)
        self.node[index].expand(candidates, tentative_policy(len(candidates)))
        self.num_nodes += 1
        return index

    def process_mini_batch(self, board: GoBoard, use_logit: bool = False):
        """tree.py:273-315."""
        queue = self.batch_queue
        planes = torch.Tensor(np.array(queue.input_plane))
        if use_logit:
            raw_policy, value_data = self.network.inference_with_policy_logits(planes)
        else:
            raw_policy, value_data = self.network.inference(planes)
        self.batch_log.append(len(queue.node_index))
        if self.eval_hook is not None:
            self.eval_hook(planes, raw_policy, value_data, use_logit)
        n_points = board.get_board_size() ** 2
        for policy, value_dist, path, node_index in zip(raw_policy, value_data, queue.path,
                                                        queue.node_index):
            by_pos = {pos: policy[i] for i, pos in enumerate(board.onboard_pos)}
            by_pos[PASS] = policy[n_points]
            if use_logit:
                by_pos[PASS] = by_pos[PASS] - 0.5
            node = self.node[node_index]
            node.update_policy(by_pos)
            node.raw_value = value_dist[1] * 0.5 + value_dist[2]
            if path:
                value = value_dist[0] + value_dist[1] * 0.5
                leaf_node, leaf_edge = path[-1]
                self.node[leaf_node].children_value[leaf_edge] = value
                for index, edge in reversed(path):
                    self.node[index].update_child_value(edge, value)
                    self.node[index].update_node_value(value)
                    value = 1.0 - value
        queue.clear()

    # ---- PUCT --------------------------------------------------------------------------
    def _initialize_search(self, board: GoBoard, color: int):
        """tree.py:49-54: tree is rebuilt from scratch every move."""
        self.num_nodes = 0
        self.current_root = self.expand_node(board, color)
        self.batch_queue.push(generate_input_planes(board, color), [], self.current_root)
        self.process_mini_batch(board)

    def search_best_move(self, board: GoBoard, color: int, time_manager: TimeManager) -> int:
        """tree.py:57-105 without the console output."""
        self._initialize_search(board, color)
        time_manager.start_timer()
        root = self.node[self.current_root]
        if root.num_children == 1:
            return PASS
        self.search(board, color, time_manager)
        if len(self.batch_queue.node_index) > 0:
            self.process_mini_batch(board)
        best = root.best_move_index()
        if root.value_evaluation(best) < RESIGN_THRESHOLD:
            return RESIGN
        return root.action[best]

    def search(self, board: GoBoard, color: int, time_manager: TimeManager):
        """tree.py:130-152."""
        search_board = board.clone()
        threshold = time_manager.get_num_visits_threshold(color)
        for _ in range(threshold):
            search_board.copy_from(board)
            self.search_mcts(search_board, color, self.current_root, [])
            if time_manager.is_time_over() or \
                    time_manager.is_move_decided(self.node[self.current_root], threshold):
                break

    def search_mcts(self, board: GoBoard, color: int, current_index: int, path: list):
        """tree.py:199-244."""
        node = self.node[current_index]
        edge = node.select_next_action(self.cgos_mode)
        move = node.action[edge]
        path.append((current_index, edge))
        board.put_stone(move, color)
        color = opponent(color)
        node.add_virtual_loss(edge)

        expand_threshold = 1
        if board.moves > 2:                                    # tree.py:224-229
            if board.record_pos(board.moves - 1) == PASS and \
                    board.record_pos(board.moves - 2) == PASS:
                expand_threshold = 10000000

        if node.children_visits[edge] + node.children_virtual_loss[edge] < expand_threshold + 1:
            if node.children_index[edge] == NOT_EXPANDED:
                child = self.expand_node(board, color)
                node.children_index[edge] = child
            else:
                child = int(node.children_index[edge])
            self.batch_queue.push(generate_input_planes(board, color), path, child)
            if len(self.batch_queue.node_index) >= self.batch_size:
                self.process_mini_batch(board)
        else:
            self.search_mcts(board, color, int(node.children_index[edge]), path)

    # ---- Gumbel / sequential halving -----------------------------------------------------
    def generate_move_with_sequential_halving(self, board: GoBoard, color: int,
                                              time_manager: TimeManager,
                                              never_resign: bool) -> int:
        """tree.py:318-356."""
        self.num_nodes = 0
        self.current_root = self.expand_node(board, color)
        self.batch_queue.push(generate_input_planes(board, color), [], self.current_root)
        self.process_mini_batch(board, use_logit=True)
        root = self.node[self.current_root]
        root.set_gumbel_noise()
        self.search_by_sequential_halving(board, color,
                                          time_manager.get_num_visits_threshold(color))
        best = root.select_root_by_halving(PLAYOUTS)
        value = root.value_evaluation(best)
        if not never_resign and value < 0.05:
            return RESIGN
        return root.action[best]

    def search_by_sequential_halving(self, board: GoBoard, color: int, threshold: int):
        """tree.py:359-384: one NN batch per phase; the count threshold restarts at 1 in
        every phase."""
        search_board = board.clone()
        n_root = self.node[self.current_root].num_children
        base = n_root if n_root < MAX_CONSIDERED_NODES else MAX_CONSIDERED_NODES
        schedule = candidates_and_visit_pairs(base, threshold)
        for num_considered, max_count in schedule.items():
            for count_threshold in range(max_count):
                for _ in range(num_considered):
                    search_board.copy_from(board)
                    self.search_sequential_halving(search_board, color, self.current_root, [],
                                                   count_threshold + 1)
            self.process_mini_batch(search_board, use_logit=True)

    def search_sequential_halving(self, board: GoBoard, color: int, current_index: int,
                                  path: list, count_threshold: int):
        """tree.py:387-422.  Leaves are queued with the child's *current* index, which is
        still NOT_EXPANDED (-1): the NN policy / raw value land in node[-1] (reference
        quirk, reproduced)."""
        node = self.node[current_index]
        if current_index == self.current_root:
            edge = node.select_root_by_halving(count_threshold)
        else:
            edge = node.select_node_by_halving()
        move = node.action[edge]
        path.append((current_index, edge))
        board.put_stone(move, color)
        color = opponent(color)
        node.add_virtual_loss(edge)
        if node.children_visits[edge] < 1:
            self.batch_queue.push(generate_input_planes(board, color), path,
                                  int(node.children_index[edge]))
        else:
            if node.children_index[edge] == NOT_EXPANDED:
                node.children_index[edge] = self.expand_node(board, color)
            self.search_sequential_halving(board, color, int(node.children_index[edge]), path,
                                           count_threshold)

    def get_root(self) -> Node:
        return self.node[self.current_root]
