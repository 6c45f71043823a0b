"""Leaf queue container with the surface of mcts/batch_data.py:7-34 (three parallel lists
`input_plane`, `path`, `node_index`, `push`, `clear`).  The GPU search keeps its queue on the
device ([tree][slot] arrays in tamago_amd/csrc/search.hip); this class is the host-visible
snapshot MCTSTree.batch_queue exposes for API compatibility."""
from typing import List, Tuple

import numpy as np


class BatchQueue:
    __slots__ = ("_leaves",)

    def __init__(self):
        self._leaves: List[Tuple[np.ndarray, List[Tuple[int, int]], int]] = []

    def push(self, input_plane: np.ndarray, path: List[Tuple[int, int]], node_index: int) -> None:
        self._leaves.append((input_plane, path, node_index))

    def clear(self) -> None:
        self._leaves = []

    def __len__(self) -> int:
        return len(self._leaves)

    @property
    def input_plane(self) -> List[np.ndarray]:
        return [leaf[0] for leaf in self._leaves]

    @property
    def path(self) -> List[List[Tuple[int, int]]]:
        return [leaf[1] for leaf in self._leaves]

    @property
    def node_index(self) -> List[int]:
        return [leaf[2] for leaf in self._leaves]
