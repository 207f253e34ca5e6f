"""Deterministic parameter generator shared by oracle/gen_golden.py and the tests (TEST INFRASTRUCTURE).

Golden fixtures store a `param_seed` instead of megabytes of weights: both the generator script (which loads the
values into the reference model) and the tests (which load them into our engine) call seeded_state().
"""
from __future__ import annotations

import numpy as np


def seeded_state(shapes, seed: int):
    """shapes: iterable of (name, shape). Returns {name: float32 array}; fan-in scaled normals, small biases."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shape in shapes:
        shape = tuple(int(s) for s in shape)
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            w = rng.standard_normal(shape) * (1.0 / np.sqrt(fan_in))
        else:
            w = rng.standard_normal(shape) * 0.05
        out[name] = w.astype(np.float32)
    return out
