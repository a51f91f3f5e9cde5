"""Region graph of a RAT-SPN (host-side structure; interface of deeprob/utils/region.py:10-99).

The layout produced here decides the gather masks the leaf kernel consumes, so for a given seed it
has to reproduce the reference bit for bit: the same ``RandomState.permutation`` call per region, in
the same order (region.py:67-74), halves sorted, repetitions concatenated level by level (:92-97).
"""
from typing import List, Optional

import numpy as np

from deeprob.utils.random import RandomState, check_random_state


class RegionGraph:
    def __init__(self, n_features: int, depth: int, random_state: Optional[RandomState] = None):
        """
        :param n_features: number of random variables.
        :param depth: number of recursive binary splits.
        :param random_state: None, an integer seed or a NumPy RandomState.
        :raises ValueError: if a parameter is out of domain (reference: region.py:42-47).
        """
        if n_features <= 0:
            raise ValueError("The number of features must be positive")
        if depth <= 0:
            raise ValueError("The region graph depth must be positive")
        if depth > int(np.log2(n_features)):
            raise ValueError("Invalid region graph depth based on the number of features")
        self.items = tuple(range(n_features))
        self.depth = depth
        self.random_state = check_random_state(random_state)

    def random_layers(self) -> List[List[tuple]]:
        """One repetition: ``[root], [partitions], [regions], [partitions], [regions], ...``."""
        layers = [[self.items]]
        for _ in range(self.depth):
            parents = layers[-1]
            children, splits = [], []
            for region in parents:
                shuffled = self.random_state.permutation(region).tolist()
                half = len(region) // 2
                left, right = tuple(sorted(shuffled[:half])), tuple(sorted(shuffled[half:]))
                children += [left, right]
                splits.append((left, right))
            layers.append(splits)
            layers.append(children)
        return layers

    def make_layers(self, n_repetitions: int = 1) -> List[List[tuple]]:
        """Concatenate ``n_repetitions`` random repetitions level by level.

        :raises ValueError: if the number of repetitions is not positive.
        """
        if n_repetitions <= 0:
            raise ValueError("n_repetitions must be at least 1")
        merged: List[List[tuple]] = [[self.items]] + [[] for _ in range(2 * self.depth)]
        for _ in range(n_repetitions):
            rep = self.random_layers()
            for level in range(1, len(rep)):
                merged[level] = merged[level] + rep[level]
        return merged
