"""Leaf queue container (mirror of mcts/batch_data.py:7-34).  The GPU search keeps its
queue on the device ([tree][slot] arrays in tamago_amd/csrc/search.hip); this class is
the host-visible snapshot MCTSTree.batch_queue exposes for API compatibility."""
from typing import List, Tuple

import numpy as np


class BatchQueue:
    def __init__(self):
        self.input_plane = []
        self.path = []
        self.node_index = []

    def push(self, input_plane: np.ndarray, path: List[Tuple[int, int]], node_index: int):
        self.input_plane.append(input_plane)
        self.path.append(path)
        self.node_index.append(node_index)

    def clear(self):
        self.input_plane = []
        self.path = []
        self.node_index = []
