"""Read-side view of one search-tree node (mirror of the MCTSNode fields and getters that
callers read after a search, mcts/node.py:21-39,159-375).  The data lives in the GPU node
pool; a view is a host snapshot fetched through ``tg_search_read_node``."""
import json

import numpy as np

from tamago_amd.mcts.constant import C_VISIT, C_SCALE


def apply_softmax(logits: np.ndarray) -> np.ndarray:
    """nn/utility.py:125-136."""
    shifted = np.exp(logits - np.max(logits))
    return shifted / np.sum(shifted)


class MCTSNode:
    def __init__(self, num_actions: int):
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

    def get_num_children(self) -> int:
        return self.num_children

    def get_best_move_index(self) -> int:
        return int(np.argmax(self.children_visits[:self.num_children]))

    def get_best_move(self) -> int:
        return self.action[self.get_best_move_index()]

    def get_child_move(self, index: int) -> int:
        return self.action[index]

    def get_child_index(self, index: int) -> int:
        return int(self.children_index[index])

    def calculate_value_evaluation(self, index: int) -> float:
        if self.children_visits[index] == 0:
            return 0.5
        return self.children_value_sum[index] / self.children_visits[index]

    def calculate_completed_q_value(self) -> np.ndarray:
        """node.py:281-305."""
        n = self.num_children
        policy = apply_softmax(self.children_policy[:n])
        q_value = np.divide(self.children_value_sum, self.children_visits,
                            out=np.zeros_like(self.children_value_sum),
                            where=(self.children_visits > 0))[:n]
        sum_prob = np.sum(policy)
        v_pi = np.sum(policy * q_value)
        value = (float(self.raw_value) * np.ones(n) + self.node_visits * v_pi / sum_prob) \
            / (self.node_visits + 1.0)
        return np.where(self.children_visits[:n] > 0, q_value, value)

    def calculate_improved_policy(self) -> np.ndarray:
        """node.py:308-321 (used by the self-play record, sgf/selfplay_record.py:56)."""
        max_visit = np.max(self.children_visits)
        sigma_base = (C_VISIT + max_visit) * C_SCALE
        logits = self.children_policy[:self.num_children] \
            + sigma_base * self.calculate_completed_q_value()
        return apply_softmax(logits)

    def select_move_by_sequential_halving_for_root(self, count_threshold: int) -> int:
        """node.py:324-346 (host copy, used for the final move choice, tree.py:344)."""
        n = self.num_children
        max_count = max(self.children_visits[:n])
        sigma_base = (C_VISIT + max_count) * C_SCALE
        counts = self.children_visits[:n] + self.children_virtual_loss[:n]
        q_mean = np.divide(self.children_value_sum, self.children_visits,
                           out=np.zeros_like(self.children_value_sum),
                           where=(self.children_visits > 0))[:n]
        evaluation = np.where(counts >= count_threshold, -10000.0,
                              self.children_policy[:n] + self.noise[:n] + sigma_base * q_mean)
        return int(np.argmax(evaluation))

    # ---- analysis strings for the GTP front end (next tier) --------------------
    def get_analysis_status_list(self, board, pv_lists_func):
        """node.py:414-450: children sorted by (visits, index) descending, unvisited ones dropped."""
        order = sorted(((int(self.children_visits[i]), i) for i in range(self.num_children)),
                       reverse=True)
        coordinate = board.coordinate
        pv_lists = pv_lists_func(self, coordinate)
        out = []
        for rank, (visits, i) in enumerate((v, i) for v, i in order if v != 0):
            move = coordinate.convert_to_gtp_format(self.action[i])
            winrate = self.children_value_sum[i] / visits
            out.append({"move": move, "visits": int(visits), "winrate": float(winrate),
                        "prior": float(self.children_policy[i]), "lcb": float(winrate),
                        "order": rank, "pv": " ".join(str(p) for p in pv_lists[move])})
        return out

    def get_analysis_from_status_list(self, mode, children_status_list):
        """node.py:453-482: "lz" = lz-analyze info lines, "cgos" = JSON."""
        if mode == "cgos":
            return json.dumps({"winrate": float(self.node_value_sum) / self.node_visits,
                               "visits": self.node_visits, "moves": children_status_list},
                              indent=None, separators=(',', ':')) + "\n"
        parts = []
        for st in children_status_list:
            if mode == "lz":
                parts.append(f"info move {st['move']} visits {st['visits']} "
                             f"winrate {int(10000 * st['winrate'])} prior {int(10000 * st['prior'])} "
                             f"lcb {int(10000 * st['lcb'])} order {st['order']} pv {st['pv']}")
        return " ".join(parts) + "\n"

    def get_analysis(self, board, mode, pv_lists_func):
        """node.py:399-411."""
        return self.get_analysis_from_status_list(mode, self.get_analysis_status_list(board, pv_lists_func))
