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
    """members of the action distribution as the native loss / sampler kernels take them: [n] for Discrete(n); one entry
    per member of a Tuple — n for Discrete(n), -D for Box(D) (2 D parameters [means | log_std], D action columns;
    action_distributions.py:197-287 composes any members) — at most 8; [] for a bare Box"""
    if is_discrete(action_space):
        return [int(action_space.n)]
    if is_tuple(action_space):
        if len(action_space.spaces) > 8:
            raise NotImplementedError("at most 8 action heads")
        out = []
        for sp in action_space.spaces:
            if is_discrete(sp):
                out.append(int(sp.n))
            elif is_box(sp):
                if len(sp.shape) != 1:
                    raise Exception("Non-trivial shape Box action spaces not currently supported. Try to flatten the space.")
                out.append(-int(sp.shape[0]))
            else:
                raise NotImplementedError(f"Tuple member {sp!r}: only Discrete and Box members are supported")
        return out
    return []


def heads_action_cols(heads) -> int:
    """action columns of a head list: one per Discrete member, D per Box(D) member (entry -D)"""
    return sum(1 if h > 0 else -h for h in heads)


def heads_mixed(heads) -> bool:
    """a Tuple with at least one Box member: its actions travel as f32 rows and are split per member for the env"""
    return any(h < 0 for h in heads)


def split_tuple_actions(rows, heads, batched: bool = True):
    """what `preprocess_actions` hands the env for a Tuple with a Box member (batched_sampling.py:51-59): one array per
    member — int32 with the action axis squeezed for a Discrete member, f32 [agents, D] for a Box(D) member.  `rows`:
    [agents, heads_action_cols] f32 (numpy or torch)."""
    out, c = [], 0
    is_torch = hasattr(rows, "dim")
    for h in heads:
        if h > 0:
            col = rows[:, c]
            if is_torch:
                import torch
                col = col.to(torch.int32)
            else:
                col = col.astype(np.int32)
            out.append(col if batched else int(col[0]))
            c += 1
        else:
            blk = rows[:, c:c - h]
            out.append(blk if batched else blk[0])
            c -= h
    return out


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
