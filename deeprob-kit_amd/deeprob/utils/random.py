"""Random-state helper (interface of deeprob/utils/random.py:8-27 in the reference)."""
from typing import Optional, Union

import numpy as np

#: Either an integer seed or a NumPy legacy RandomState.
RandomState = Union[int, np.random.RandomState]


def check_random_state(random_state: Optional[RandomState] = None) -> np.random.RandomState:
    """Normalise ``random_state`` to a ``np.random.RandomState``.

    None -> a fresh unseeded generator, int -> seeded generator, RandomState -> itself.
    :raises ValueError: for anything else (reference: deeprob/utils/random.py:21-27).
    """
    if isinstance(random_state, np.random.RandomState):
        return random_state
    if random_state is None:
        return np.random.RandomState()
    if isinstance(random_state, int):
        return np.random.RandomState(random_state)
    raise ValueError("The random state must be either None, a seed integer or a Numpy RandomState object")
