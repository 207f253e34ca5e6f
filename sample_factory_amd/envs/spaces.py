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

    def __repr__(self):
        return "Dict(" + ", ".join(f"{k!r}: {v!r}" for k, v in self.spaces.items()) + ")"


class Tuple:
    """gymnasium.spaces.Tuple stand-in (multi-head action spaces, e.g. VizDoom: move / turn / attack)"""

    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __iter__(self):
        return iter(self.spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __repr__(self):
        return f"Tuple({', '.join(repr(s) for s in self.spaces)})"


def is_discrete(space) -> bool:
    return hasattr(space, "n")


def is_tuple(space) -> bool:
    return hasattr(space, "spaces") and not hasattr(space, "keys")


def action_head_sizes(action_space):
    """sizes of the categorical heads: [n] for Discrete(n), [n1, n2, ...] for a Tuple of Discrete spaces (the only
    Tuple the native loss / sampler kernels take; a Tuple containing a Box is rejected)"""
    if is_discrete(action_space):
        return [int(action_space.n)]
    if is_tuple(action_space):
        if not all(is_discrete(sp) for sp in action_space.spaces):
            raise NotImplementedError("Tuple action spaces are supported for Discrete members only")
        if len(action_space.spaces) > 8:
            raise NotImplementedError("at most 8 action heads")
        return [int(sp.n) for sp in action_space.spaces]
    return []


def is_box(space) -> bool:
    return not hasattr(space, "n") and hasattr(space, "shape") and getattr(space, "shape", None) is not None and not hasattr(space, "spaces")


def calc_num_actions(action_space) -> int:
    """action_distributions.py:14-26"""
    if is_discrete(action_space):
        return 1
    if is_tuple(action_space):
        return sum(calc_num_actions(a) for a in action_space.spaces)
    if is_box(action_space):
        if len(action_space.shape) != 1:
            raise Exception("Non-trivial shape Box action spaces not currently supported. Try to flatten the space.")
        return action_space.shape[0]
    raise NotImplementedError(f"Action space type {type(action_space)} not supported!")


def calc_num_action_parameters(action_space) -> int:
    """action_distributions.py:29-38"""
    if is_discrete(action_space):
        return action_space.n
    if is_tuple(action_space):
        return sum(calc_num_action_parameters(a) for a in action_space.spaces)
    if is_box(action_space):
        return int(np.prod(action_space.shape)) * 2
    raise NotImplementedError(f"Action space type {type(action_space)} not supported!")
