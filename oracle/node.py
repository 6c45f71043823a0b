"""Oracle MCTS node math (TEST INFRASTRUCTURE - see oracle/__init__.py).

Restates mcts/node.py:21-375 and mcts/pucb/pucb.py:8-29.  Array dtypes follow
node.py:27-39 exactly, and leaf values flow through as 0-d float32 torch tensors as
they do in the reference (mcts/tree.py:297-313), because that decides the arithmetic:
``children_value_sum[e] += value`` is executed by torch as a FLOAT32 addition whose
result is stored back into the float64 array, ``1.0 - value`` is float32, and
``node_value_sum`` turns into a float32 tensor (verified in the build container with
numpy 2.2.6 / torch 2.10).
"""
import math

import numpy as np

NOT_EXPANDED = -1          # mcts/constant.py:5
C_VISIT = 50               # mcts/constant.py:17
C_SCALE = 1.0              # mcts/constant.py:20
PUCB_SECOND_TERM_WEIGHT = 1.0  # mcts/constant.py:8


def apply_softmax(logits: np.ndarray) -> np.ndarray:
    """nn/utility.py:125-136."""
    e = np.exp(logits - np.max(logits))
    return e / np.sum(e)


def pucb_values(node_visits: int, child_counts: np.ndarray, value_sum: np.ndarray,
                policy: np.ndarray) -> np.ndarray:
    """pucb.py:8-29: Q = value_sum / n (0 where n == 0) plus
    policy * sqrt(N + 1) / (n + 1), all float64, math.sqrt for the root term."""
    q = np.divide(value_sum, child_counts, out=np.zeros_like(value_sum),
                  where=(child_counts != 0))
    u = PUCB_SECOND_TERM_WEIGHT * policy * math.sqrt(node_visits + 1) / (child_counts + 1)
    return q + u


class Node:
    """node.py:21-39."""

    def __init__(self, num_actions: int):
        self.num_actions = num_actions
        self.node_visits = 0
        self.virtual_loss = 0
        self.node_value_sum = 0.0
        self.raw_value = 0.0
        self.action = [0] * num_actions
        self.children_index = np.zeros(num_actions, dtype=np.int32)
        self.children_value = np.zeros(num_actions, dtype=np.float64)
        self.children_visits = np.zeros(num_actions, dtype=np.int32)
        self.children_policy = np.zeros(num_actions, dtype=np.float64)
        self.children_virtual_loss = np.zeros(num_actions, dtype=np.int32)
        self.children_value_sum = np.zeros(num_actions, dtype=np.float64)
        self.noise = np.zeros(num_actions, dtype=np.float64)
        self.num_children = 0

    def expand(self, candidates, prior):
        """node.py:41-73 (expand + set_policy).  children_policy beyond num_children
        keeps whatever the slot held before, as in the reference (never read)."""
        self.node_visits = 0
        self.virtual_loss = 0
        self.node_value_sum = 0.0
        self.raw_value = 0.0
        self.action = [0] * self.num_actions
        self.children_index.fill(NOT_EXPANDED)
        self.children_value.fill(0.0)
        self.children_visits.fill(0)
        self.children_virtual_loss.fill(0)
        self.children_value_sum.fill(0.0)
        self.noise.fill(0.0)
        for i, (pos, p) in enumerate(zip(candidates, prior)):
            self.action[i] = pos
            self.children_policy[i] = p
        self.num_children = len(candidates)

    def add_virtual_loss(self, index: int):
        """node.py:76-83."""
        self.virtual_loss += 1
        self.children_virtual_loss[index] += 1

    def update_policy(self, policy_by_pos):
        """node.py:86-93: no masking, no renormalisation."""
        for i in range(self.num_children):
            self.children_policy[i] = policy_by_pos[self.action[i]]

    def update_child_value(self, index: int, value):
        """node.py:118-127."""
        self.children_value_sum[index] += value
        self.children_visits[index] += 1
        self.children_virtual_loss[index] -= 1

    def update_node_value(self, value):
        """node.py:130-138."""
        self.node_value_sum += value
        self.node_visits += 1
        self.virtual_loss -= 1

    def select_next_action(self, cgos_mode: bool) -> int:
        """node.py:141-157."""
        scores = pucb_values(self.node_visits + self.virtual_loss,
                             self.children_visits + self.children_virtual_loss,
                             self.children_value_sum, self.children_policy + self.noise)
        if cgos_mode:
            scores[self.num_children - 1] -= 0.1
        return int(np.argmax(scores[:self.num_children]))

    def best_move_index(self) -> int:
        """node.py:169-176."""
        return int(np.argmax(self.children_visits[:self.num_children]))

    def value_evaluation(self, index: int) -> float:
        """node.py:364-375."""
        if self.children_visits[index] == 0:
            return 0.5
        return self.children_value_sum[index] / self.children_visits[index]

    # ---- Gumbel / sequential halving --------------------------------------------------
    def set_gumbel_noise(self):
        """node.py:275-278: replaces the array; size A, not num_children."""
        self.noise = np.random.gumbel(loc=0.0, scale=1.0, size=self.noise.size)

    def completed_q(self) -> np.ndarray:
        """node.py:281-305 (mixed value approximation)."""
        n = self.num_children
        pi = apply_softmax(self.children_policy[:n])
        q = np.divide(self.children_value_sum, self.children_visits,
                      out=np.zeros_like(self.children_value_sum),
                      where=(self.children_visits > 0))[:n]
        sum_prob = np.sum(pi)
        v_pi = np.sum(pi * q)
        raw = float(self.raw_value)              # float32 -> float64 is exact
        mixed = (raw * np.ones(n, dtype=np.float64) + self.node_visits * v_pi / sum_prob) \
            / (self.node_visits + 1.0)
        return np.where(self.children_visits[:n] > 0, q, mixed)

    def improved_policy(self) -> np.ndarray:
        """node.py:308-321.  max over ALL A slots, as in the reference."""
        max_visit = np.max(self.children_visits)
        sigma = (C_VISIT + max_visit) * C_SCALE
        return apply_softmax(self.children_policy[:self.num_children] + sigma * self.completed_q())

    def select_root_by_halving(self, count_threshold: int) -> int:
        """node.py:324-346."""
        n = self.num_children
        max_count = max(self.children_visits[:n])
        sigma = (C_VISIT + max_count) * C_SCALE
        counts = self.children_visits[:n] + self.children_virtual_loss[:n]
        q = np.divide(self.children_value_sum, self.children_visits,
                      out=np.zeros_like(self.children_value_sum),
                      where=(self.children_visits > 0))[:n]
        score = np.where(counts >= count_threshold, -10000.0,
                         self.children_policy[:n] + self.noise[:n] + sigma * q)
        return int(np.argmax(score))

    def select_node_by_halving(self) -> int:
        """node.py:349-361."""
        n = self.num_children
        score = self.improved_policy() - (self.children_visits[:n] / (1.0 + self.node_visits))
        return int(np.argmax(score))
