"""Minimal observation/action space descriptors.

gymnasium is the reference's dependency for these (envs/env_utils.py, algo/utils/action_distributions.py:14-38); it
is not installed on the GPU boxes, so the engine duck-types: anything with `.n` is Discrete, anything with `.shape`
and `.dtype` is a Box, anything with `.spaces` mapping is a Dict.  Real gymnasium spaces work unchanged.
"""
from __future__ import annotations

import numpy as np


class Discrete:
    def __init__(self, n: int):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def __repr__(self):
        return f"Discrete({self.n})"


class Box:
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high = low, high
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)

    def __repr__(self):
        return f"Box({self.shape}, {self.dtype})"


class Dict:
    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def keys(self):
        return self.spaces.keys()

    def items(self):
        return self.spaces.items()

    def __getitem__(self, k):
        return self.spaces[k]


def is_discrete(space) -> bool:
    return hasattr(space, "n")


def is_box(space) -> bool:
    return not hasattr(space, "n") and hasattr(space, "shape") and getattr(space, "shape", None) is not None and not hasattr(space, "spaces")


def calc_num_actions(action_space) -> int:
    """action_distributions.py:14-26"""
    if is_discrete(action_space):
        return 1
    if is_box(action_space):
        if len(action_space.shape) != 1:
            raise Exception("Non-trivial shape Box action spaces not currently supported. Try to flatten the space.")
        return action_space.shape[0]
    raise NotImplementedError(f"Action space type {type(action_space)} not supported!")


def calc_num_action_parameters(action_space) -> int:
    """action_distributions.py:29-38"""
    if is_discrete(action_space):
        return action_space.n
    if is_box(action_space):
        return int(np.prod(action_space.shape)) * 2
    raise NotImplementedError(f"Action space type {type(action_space)} not supported!")
