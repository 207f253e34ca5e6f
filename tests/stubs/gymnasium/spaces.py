"""gymnasium.spaces stand-in: the engine's own duck-typed descriptors under gymnasium's names (see __init__.py)"""
import numpy as np

from sample_factory_amd.envs import spaces as _sp


class Space:
    pass


class Box(_sp.Box, Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        if shape is None:
            shape = np.shape(low)
        _sp.Box.__init__(self, low, high, shape, dtype)


class Discrete(_sp.Discrete, Space):
    def __init__(self, n, seed=None, start=0):
        _sp.Discrete.__init__(self, n)


class Dict(_sp.Dict, Space):
    def __init__(self, spaces=None, seed=None, **kw):
        _sp.Dict.__init__(self, dict(spaces or {}, **kw))


class Tuple(_sp.Tuple, Space):
    def __init__(self, spaces, seed=None):
        _sp.Tuple.__init__(self, spaces)
